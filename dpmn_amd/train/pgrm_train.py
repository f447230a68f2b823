"""Training-mode PGRM: explicit forward (keeps the activations the backward needs) and explicit backward, both
composed from libdpmn_hip.so kernels.  Replaces autograd through model/pgrm.py (loss.backward(),
interfaces/super_resolution.py:270) for PGRM.forward (pgrm.py:546-565).

The forward here is the unfused variant of csrc/pgrm_forward.hip (LayerNorm, GELU kept as separate kernels so that
pre-activations are available).

Train-mode Dropout / attn_drop / DropPath (pgrm.py:32,40,248,329-330,554-555): masks are counter-based (a pure function of
a 64-bit seed and the element index, include/dpmn_hip.h), regenerated in the backward instead of stored.  The seeds of one
PGRM call are drawn from torch's CPU generator (draw_seeds), so torch.manual_seed makes a run reproducible; the reference's
Philox masks themselves cannot be replayed by any other implementation, parity there is distributional.
"""
import ctypes as C
import os

import torch
from .._abi import stream_of as _abi_stream_of

from .. import _abi, ops
from .._abi import dptr, lib, check, stream
from ..model import packing

GELU = ops.ACT["gelu"]


_ZB = {}


def _zero_bias(n, device):
    """a shared all-zero bias vector (never written)"""
    z = _ZB.get((n, device))
    if z is None:
        z = _ZB[(n, device)] = torch.zeros(n, device=device)
    return z


def _e(*shape, like):
    return torch.empty(*shape, device=like.device)


def _z(*shape, like):
    return torch.zeros(*shape, device=like.device)


def layernorm(x, w, b):
    y = torch.empty_like(x)
    check(lib.dpmn_layernorm_f32(dptr(x), dptr(w), dptr(b), 1e-5, dptr(y), x.shape[0], x.shape[1], stream()))
    return y


# 1 (default): dgamma / dbeta as per-block partials added in block order by ONE k_tn_reduce launch (dpmn_layernorm_bwd_det_f32:
# bitwise reproducible; measured the same step time as the atomics -- with two finish launches per call it had cost +0.6 ms per
# step: every tiny serial launch of the two-stream PGRM backward is paid in wall time); 0: the atomics
LNB_DET = os.environ.get("DPMN_LNB_DET", "1") != "0"
# depthwise-conv weight / bias and pointwise-conv bias gradients as per-image partial rows added in image order (no atomics)
DET_SMALL = os.environ.get("DPMN_DET_SMALL", "1") != "0"


def layernorm_bwd(x, dy, w, dx, accumulate, dgamma, dbeta):
    """LayerNorm backward (dx, dgamma +=, dbeta +=); LNB_DET selects the atomics-free form."""
    if not LNB_DET:
        check(lib.dpmn_layernorm_bwd_f32(dptr(x), dptr(dy), dptr(w), 1e-5, dptr(dx), int(accumulate), dptr(dgamma), dptr(dbeta),
                                         x.shape[0], x.shape[1], stream()))
        return
    ptr, nb = _ws(x.device, 512 * 2 * x.shape[1] * 4)
    check(lib.dpmn_layernorm_bwd_det_f32(dptr(x), dptr(dy), dptr(w), 1e-5, dptr(dx), int(accumulate), dptr(dgamma), dptr(dbeta),
                                         x.shape[0], x.shape[1], ptr, nb, stream()))
    _ws_done()


def act_fwd(x, act=GELU):
    y = torch.empty_like(x)
    check(lib.dpmn_act_fwd_f32(dptr(x), dptr(y), act, 0.0, x.numel(), stream()))
    return y


def act_bwd(dy, pre, act=GELU):
    d = torch.empty_like(pre)
    check(lib.dpmn_act_bwd_f32(dptr(dy), dptr(pre), dptr(d), act, 0.0, pre.numel(), stream()))
    return d


TN_DEFER = os.environ.get("DPMN_TN_DEFER", "1") != "0"
_TN_WS = {}
_tn_pending = None     # set by PGRMFunction.backward for one PGRM backward: [arena tensor, bytes used, keep-alive tensors]


def _tn_workspace(device):
    """Arena of one PGRM backward (per device and stream): every partial-row buffer of its atomics-free reductions -- the split
    partials of the 14 Linear weight gradients (~120 MB at B = 48), the pointwise-conv weight gradient's 32 splits (19 MB per
    block), column / row sums, LayerNorm parameter gradients -- lives here until the backward's ONE reduce launch."""
    key = (device.type, device.index, _abi_stream_of(device))
    if key not in _TN_WS:
        _TN_WS[key] = torch.empty(80 << 20, device=device)      # 320 MB
    return _TN_WS[key]


def _ws(device, nbytes):
    """(pointer, bytes) of a workspace slice for one atomics-free reduction: a slice of the backward's arena while its reductions
    are deferred (include/dpmn_hip.h dpmn_reduce_defer_*: the slice must stay untouched until the flush), else the shared
    per-stream scratch (the reduction then runs at once)."""
    if _tn_pending is not None:
        nbytes = (int(nbytes) + 255) // 256 * 256
        arena = _tn_pending[0]
        if nbytes <= arena.numel() * 4:
            if _tn_pending[1] + nbytes > arena.numel() * 4:
                tn_flush()
            ptr = arena.data_ptr() + _tn_pending[1]
            _tn_pending[1] += nbytes
            return ptr, nbytes
        check(lib.dpmn_reduce_defer_enable(0))       # does not fit the arena at all: run this one immediately
        _tn_pending[2].append("resume")
    ws = ops.splitk_workspace(device)
    return ws.data_ptr(), ws.numel() * 4


def _ws_done():
    if _tn_pending is not None and _tn_pending[2] and _tn_pending[2][-1] == "resume":
        _tn_pending[2].pop()
        check(lib.dpmn_reduce_defer_enable(1))


def tn_flush(end=False):
    """One multi-descriptor reduce launch for every queued ordered reduction (dpmn_reduce_defer_flush); the arena is free again."""
    if _tn_pending is None:
        return
    check(lib.dpmn_reduce_defer_flush(1 if end else 0, stream()))
    _tn_pending[1] = 0
    del _tn_pending[2][:]


def gemm_tn(dy, x, dw, db=None, leaf=True):
    """dw (N,K) += dy (M,N)^T x (M,K) ; db (N) += column sums of dy (fused into the same kernel).
    Inside a PGRM backward (leaf=True: nothing reads dw / db before its end) only the split partial sums are launched, each into
    its own slice of the backward's arena; ONE reduce launch at the end adds them all (tn_flush) -- the two-stream phase of the
    backward is launch-rate sensitive.  leaf=False: the result is read right away, the reduction runs immediately."""
    M, N, K = dy.shape[0], dy.shape[1], x.shape[1]
    if _tn_pending is not None and not leaf:
        check(lib.dpmn_reduce_defer_enable(0))
    try:
        ptr, nb = _ws(dy.device, lib.dpmn_gemm_tn_partial_bytes(M, N, K)) if leaf else (ops.splitk_workspace(dy.device).data_ptr(),
                                                                                       ops.splitk_workspace(dy.device).numel() * 4)
        check(lib.dpmn_gemm_tn_f32(dptr(dy), dptr(x), dptr(dw), dptr(db, True), M, N, K, ptr, nb, stream()))
        if leaf:
            _ws_done()
    finally:
        if _tn_pending is not None and not leaf:
            check(lib.dpmn_reduce_defer_enable(1))


def defer_rows(part, dw, db, NK, N, rows):
    """dw (NK) += sum over `rows` rows of part[:, :NK], db (N) += ... part[:, NK:] in ROW ORDER (atomics-free finish of per-block /
    per-image partial sums).  Inside a PGRM backward the sum joins the one multi-descriptor reduce launch at its end (tn_flush);
    `part` is kept alive until then."""
    if _tn_pending is not None:
        _tn_pending[2].append((part, dw, db))
    check(lib.dpmn_rows_reduce_f32(dptr(part), dptr(dw), dptr(db, True), NK, N, rows, stream()))


def colsum(dy, db):
    """db (N) += column sums of dy (M, N), without atomics (per-block partials summed in block order: bitwise reproducible)."""
    M, N = dy.shape
    ptr, nb = _ws(dy.device, (M + 255) // 256 * N * 4)
    check(lib.dpmn_colsum_det_f32(dptr(dy), dptr(db), M, N, ptr, nb, stream()))
    _ws_done()


def linear_bwd(dy, x, w, dw, db):
    """y = x w^T + b : returns dx; accumulates dw, db."""
    gemm_tn(dy, x, dw, db)
    return ops.linear(dy, packing.transposed(w))


def params_of(m):
    """list(m.parameters()), cached on the module (the module tree walk costs ~0.1 ms per call and every forward / backward asks
    three times; Parameter objects are never replaced on this path -- the Trainer only rebinds their .data)."""
    ps = m.__dict__.get("_dpmn_plist")
    if ps is None:
        ps = m.__dict__["_dpmn_plist"] = list(m.parameters())
    return ps


def fn_inputs(m):
    """What a module's autograd Function takes besides its activations: the parameters -- or, in direct mode (the backward kernels
    write the flat bucket's gradient views themselves and return None for every parameter), ONE anchor tensor that keeps the node in
    the graph: ~100 parameters per module as Function inputs cost one AccumulateGrad node each per step (569 of them, ~3 ms of the
    autograd thread's time) for gradients that are never handed to them."""
    b = getattr(m, "_dpmn_bucket", None)
    return [b.anchor] if b is not None else params_of(m)


def grad_targets(m):
    """{param: tensor the backward kernels accumulate into}, direct?  Direct mode (train/optim.py FlatBucket(direct=True)):
    the targets are the flat bucket's zeroed gradient views; otherwise freshly zeroed tensors handed back to autograd."""
    ps = params_of(m)
    if getattr(m, "_dpmn_bucket", None) is not None:
        return {p: p._dpmn_sink for p in ps}, True
    return {p: torch.zeros_like(p) for p in ps}, False


def finish_grads(m, gr, direct):
    """tuple of per-parameter gradients for autograd (None in direct mode, after signalling the bucket)."""
    if direct:
        m._dpmn_bucket.grads_ready()
        return (None,)          # the anchor (fn_inputs)
    return tuple(gr[p] for p in params_of(m))


def conv_wgrad_into(d, dy, dweight, layout="conv", phase=None):
    """Accumulate the weight gradient of the conv described by `d` straight into the parameter-layout tensor `dweight`:
    layout "conv" = nn.Conv2d (Cout,Cin,KH,KW); "convT_s1" = nn.ConvTranspose2d(stride 1) (Cin,Cout,KH,KW), flipped taps;
    "convT_s2k4" = phase (py,px) of nn.ConvTranspose2d(4,2,1) (tap ky' -> ky = (1-py)+2ky', packing.pack_convT_s2k4).
    Channels the descriptor only carries as padding (beyond dweight's real Cin / Cout) are dropped."""
    assert dweight.is_contiguous()
    if layout == "conv":
        co, ci, kh, kw = dweight.shape
        st = (ci * kh * kw, kh * kw, kw, 1, 0)
    elif layout == "convT_s1":
        ci, co, kh, kw = dweight.shape
        st = (kh * kw, co * kh * kw, -kw, -1, (kh - 1) * kw + (kw - 1))
    else:
        ci, co = dweight.shape[:2]
        st = (16, co * 16, 8, 2, (1 - phase[0]) * 4 + (1 - phase[1]))
    # Every pixel split STORES its partial (Cout, Kp) tile set into its own copy of a persistent packed workspace (no atomics:
    # deterministic, and no zero-init), then one unpack kernel sums the copies into the parameter layout.
    # (Measured alternatives: atomics into one packed copy + re-zeroing unpack, ~51 TF on the CMM convs; atomics straight into
    # the parameter layout, dpmn_conv2d_wgrad_strided_f32, 3x slower than that.)
    cin_d = sum(d.cseg[i] for i in range(3) if d.inp[i])
    kp = (d.KH * d.KW * cin_d + 31) // 32 * 32
    slots = C.c_int(0)
    if WGRAD_MODE != "atomic":
        check(lib.dpmn_conv2d_wgrad_excl_slots(C.byref(d), C.cast(C.pointer(slots), C.c_void_p)))
    if slots.value == 0:      # (the library advises the slotted-atomic path for tiny gradients over many pixels)
        nslots = 32 if d.Cout * kp <= 12288 else 1     # small gradients: spread the pixel splits' atomics over 32 copies
        key = (dy.device, _abi_stream_of(dy.device), d.Cout, kp, nslots)
        ws = _WGRAD_WS.get(key)
        if ws is None:
            ws = _WGRAD_WS[key] = torch.zeros(nslots, d.Cout, kp, device=dy.device)
        check(lib.dpmn_conv2d_wgrad_f32(C.byref(d), dptr(dy), dptr(ws), nslots, stream()))
        check(lib.dpmn_conv2d_wgrad_unpack_f32(dptr(ws), dptr(dweight), d.Cout, cin_d, d.KH, d.KW, co, ci, *st, 1, nslots, stream()))
        return
    n = slots.value * d.Cout * kp
    if UNPACK_QUEUE is not None and not torch.cuda.is_current_stream_capturing():
        # deferred: the weight gradient goes to this conv's OWN persistent slot workspace; one multi-descriptor unpack launch at
        # the end of the module's backward sums the slots of every conv into the parameter layout (UnpackQueue.flush)
        ws = UNPACK_QUEUE.region((dweight.data_ptr(), layout, phase, d.Cout, cin_d, d.KH, d.KW, slots.value), n, dweight, (d.Cout, cin_d, d.KH, d.KW, co, ci, st),
                                 slots.value)
        check(lib.dpmn_conv2d_wgrad_excl_f32(C.byref(d), dptr(dy), dptr(ws), slots.value, stream()))
        return
    skey = (dy.device, _abi_stream_of(dy.device))
    ws = _WGRAD_WS.get(skey)
    if ws is None or ws.numel() < n:       # one workspace per (device, stream): wgrad -> unpack pairs are stream-ordered
        ws = _WGRAD_WS[skey] = torch.empty(max(n, 16 << 20), device=dy.device)
    check(lib.dpmn_conv2d_wgrad_excl_f32(C.byref(d), dptr(dy), dptr(ws), slots.value, stream()))
    check(lib.dpmn_conv2d_wgrad_unpack_f32(dptr(ws), dptr(dweight), d.Cout, cin_d, d.KH, d.KW, co, ci, *st, 0, slots.value, stream()))


class UnpackQueue:
    """The weight-gradient unpacks of one module's backward (the CMM: ~40 convs, 57 launches of k_wgrad_unpack per step) as ONE launch.
    Every conv gets its own persistent exclusive-slot workspace (the wgrad kernel's partial tiles, no zero-init) and a 112-byte
    descriptor in a device table built once (the gradient sinks of a Trainer and these workspaces keep their addresses for the
    whole run); flush() sums the slots of every conv that ran since the last flush into the parameter layout."""

    def __init__(self):
        self.entries, self.order, self.touched, self.tables = {}, [], set(), {}

    def region(self, key, n, dweight, geom, slots):
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = [torch.empty(n, device=dweight.device), dweight, geom, slots]
            self.order.append(key)
        self.touched.add(key)
        return e[0]

    def flush(self):
        """One launch for every conv that ran since the last flush.  The device table is cached per SET of convs: the whole module
        (one flush per backward) or one exchange segment of it (train/cmm_train.py: a flush per segment when the gradients are
        exchanged in segments)."""
        import struct
        if not self.touched:
            return
        keys = tuple(k for k in self.order if k in self.touched)
        tab = self.tables.get(keys)
        if tab is None:
            raw, prefix, nb, nbytes = bytearray(), [], 0, 0.0
            shape = (C.c_int * 3)()
            for key in keys:
                ws, dw, (cout, cin, kh, kw, co, ci, st), slots = self.entries[key]
                nbytes += 4.0 * (ws.numel() + 2 * min(co, cout) * min(ci, cin) * kh * kw)      # every slot read once, the gradient read + written
                K = kh * kw * cin
                kp = (K + 31) // 32 * 32
                check(lib.dpmn_conv_pack_tile_shape(cout, cin, kh * kw, st[0], st[1], C.cast(shape, C.c_void_p)))
                co_t, ci_t, order = shape[0], shape[1], shape[2]
                nci = (cin + ci_t - 1) // ci_t
                raw += struct.pack("<2Q6q12i", dw.data_ptr(), ws.data_ptr(), st[0], st[1], st[2], st[3], st[4], cout * kp, cout, kp, K, cin, kw,
                                   min(co, cout), min(ci, cin), co_t, ci_t, order, nci, slots)
                prefix.append(nb)
                nb += (cout + co_t - 1) // co_t * nci
            dev = self.entries[keys[0]][0].device
            tab = self.tables[keys] = (torch.frombuffer(raw, dtype=torch.uint8).to(dev), torch.tensor(prefix, dtype=torch.int32, device=dev), nb, nbytes)
        descs, prefix, nb, nbytes = tab
        lib.dpmn_profile_hint_bytes(nbytes)
        check(lib.dpmn_conv2d_wgrad_unpack_multi_f32(descs.data_ptr(), prefix.data_ptr(), len(keys), nb, stream()))
        self.touched.clear()


UNPACK_QUEUE = None      # set by a module's backward (train/cmm_train.py) around its conv_wgrad_into calls
UNPACK_MULTI = os.environ.get("DPMN_UNPACK_MULTI", "1") != "0"

import os as _os
WGRAD_MODE = _os.environ.get("DPMN_WGRAD", "excl")      # "atomic": the previous accumulate-by-atomics path (A/B switch)
_WGRAD_WS = {}


class ConvSpec:
    """Geometry + packed weights of one conv in the reference layout (Cout, Cin, k, k), stride 1, 'same' padding."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        self.cout, self.cin, self.k = weight.shape[0], weight.shape[1], weight.shape[2]
        self.pad = (self.k - 1) // 2

    def forward(self, x):
        return ops.conv2d([x], packing.tpack_conv(self.weight), self.bias, self.cout, self.k, pad=self.pad)

    def backward(self, x, dy, dweight, dbias, need_dx=True):
        """x (B,H,W,Cin) input of the forward, dy (B,H,W,Cout); accumulates dweight/dbias; returns dx."""
        B, H, W, _ = x.shape
        d = ops.conv_desc([x], self.k, pad=self.pad, cout=self.cout)
        conv_wgrad_into(d, dy, dweight)
        colsum(dy.reshape(-1, self.cout), dbias)
        if not need_dx:
            return None
        wt = packing.tpack_dgrad_conv_s1(self.weight, 0, self.cin)   # data gradient = transposed conv with the same weight tensor
        return ops.conv2d([dy], wt, None, self.cin, self.k, pad=self.pad)


FUSED_ATTN = os.environ.get("DPMN_ATTN_FUSED_TRAIN", "1") != "0"      # 0: LayerNorm / q / kv / window-attention as separate launches

FUSED_ATTN_BWD = os.environ.get("DPMN_ATTN_FUSED_BWD", "1") != "0"    # 0: the training forward saves q / kv, the backward runs one attention kernel per window size
FUSED_SKMLP = os.environ.get("DPMN_SKMLP_TRAIN", "1") != "0"          # 0: select / proj_head / LayerNorm2 / fc1 as separate launches

N_SEEDS = 12     # [0] pos_drop(x_q) [1] pos_drop(x_kv); block bi at 2+5*bi: attn_drop, DropPath(attn), Mlp drop 1, Mlp drop 2, DropPath(mlp)


def draw_seeds():
    return torch.randint(0, 2 ** 62, (N_SEEDS,), dtype=torch.int64).tolist()


def drop_config(m):
    """None in eval / all-zero rates, else dict(p, pa, dp=(block0, block1), seeds)."""
    p, pa, dp = m.drop_probs
    if not m.training or (p <= 0 and pa <= 0 and max(dp) <= 0):
        return None
    return dict(p=p, pa=pa, dp=tuple(dp), seeds=draw_seeds())


NATIVE_FWD = os.environ.get("DPMN_PGRM_NATIVE_TRAIN", "1") != "0"     # 0: issue the training forward op by op from Python (A/B switch)


def _saved_layout(m, B):
    """Offsets (floats, 64-float aligned) of every tensor the backward reads inside one slab: [(key, block or None, offset, shape)]."""
    H, Wd = m.patches_resolution
    L, Cd, G = H * Wd, m.embed_dim, len(m.window_size)
    M, Ch = B * L, int(m.embed_dim * m.mlp_ratio)
    Cm = m.hidden_size * m.patch * m.patch
    fold = lib.dpmn_ln_qkv_window_attn_workspace_bytes() // 4
    per_block = [("cat", (M, Cd)), ("fold", (fold,)), ("feats", (M, Cd)), ("partial", (B * ((L + 31) // 32), Cd)), ("avec", (B, G, Cd // G)),
                 ("x1", (M, Cd)), ("ypre", (M, Ch)), ("V", (M, Cd // G)), ("n2", (M, Cd)), ("gpre", (M, Ch)), ("g", (M, Ch)), ("z", (M, Ch)),
                 ("tkv_out", (M, Cd))]
    items = [("tq", None, (M, Cd)), ("tkv0", None, (M, Cd))]
    for bi in range(2):
        items += [(k, bi, shp) for k, shp in per_block]
    items += [("c0", None, (B, H, Wd, Cm)), ("c1", None, (B, H, Wd, Cm))]
    out, off = [], 0
    for key, bi, shp in items:
        n = 1
        for v in shp:
            n *= v
        out.append((key, bi, off, n, shp))
        off += (n + 63) // 64 * 64
    return out, off


def _forward_native(m, x_q, x_kv, residuals, drop, w):
    """forward() below as ONE native call (csrc/pgrm_forward.hip dpmn_pgrm_forward_train_f32): same kernels, same order, same saved
    tensors -- carved out of one slab -- without the host time of ~45 ctypes calls and tensor allocations per module."""
    import ctypes as C
    B = x_kv.shape[0]
    H, Wd = m.patches_resolution
    lay = m.__dict__.setdefault("_train_layout", {})
    if B not in lay:
        lay[B] = _saved_layout(m, B)
    items, total = lay[B]
    slab = torch.empty(total, device=x_kv.device)
    base = slab.data_ptr()
    cs = _abi.PgrmSaved()
    sv = dict(x_q=x_q, x_kv=x_kv, residuals=list(residuals), blocks=[{}, {}], drop=drop)
    for key, bi, off, n, shp in items:
        t = slab.narrow(0, off, n).view(shp)
        if bi is None:
            setattr(cs, key, base + 4 * off)
            sv[key] = t
        else:
            setattr(cs.blk[bi], key, base + 4 * off)
            sv["blocks"][bi][key] = t
    for bi, blk in enumerate(m.layers[0].blocks):
        s = sv["blocks"][bi]
        s["tkv_in"] = sv["tkv0"] if bi == 0 else sv["blocks"][bi - 1]["tkv_out"]
        s["win"] = [min(H, Wd) if min(H, Wd) <= ws else ws for ws in m.window_size]
        s["shift"] = [0 if (bi == 0 or min(H, Wd) <= ws) else ws // 2 for ws in m.window_size]
        s["tables"] = [getattr(blk.attn, "relative_position_bias_table_%d" % g) for g in range(len(m.window_size))]
        s["q"] = s["kv"] = s["nq"] = s["nkv"] = None
    sv["tkv_out"] = sv["blocks"][1]["tkv_out"]
    cd = None
    if drop:
        cd = _abi.PgrmDrop()
        cd.p, cd.pa = drop["p"], drop["pa"]
        cd.dp[0], cd.dp[1] = drop["dp"]
        for i, v in enumerate(drop["seeds"]):
            cd.seeds[i] = v
    d = _abi.ConvDesc()
    ops._attach_workspace(d, x_kv.device)
    sc = _abi.CmmScratch(d.splitk_ws, d.splitk_ws_bytes, d.arrive_cnt, d.arrive_cnt_len)
    t0 = packing.tpack_conv(m.conv_before_upsample[0].weight)
    t1 = packing.tpack_conv(m.conv_before_upsample[1].weight)
    out = torch.empty(B, m.hidden_size, m.img_size[0], m.img_size[1], device=x_kv.device)
    check(lib.dpmn_pgrm_forward_train_f32(C.byref(w), dptr(x_q), x_q.shape[1], dptr(x_kv), _abi.ptr_array(residuals), len(residuals), dptr(t0),
                                          dptr(t1), C.byref(cd) if cd is not None else None, C.byref(cs), C.byref(sc), dptr(out), B, stream()))
    sv["_slab"], sv["_cs"] = slab, cs
    return out, sv


def forward(m, x_q, x_kv, residuals, drop=None):
    """Returns (out, saved).  m: dpmn_amd.model.pgrm.PGRM; drop: drop_config(m)."""
    B = x_kv.shape[0]
    H, Wd = m.patches_resolution
    L, Cd = H * Wd, m.embed_dim
    M, Ch, G = B * L, int(m.embed_dim * m.mlp_ratio), len(m.window_size)
    hpg = m.num_heads // G
    pe = m.patch_embed
    fuse = x_q.shape[1] == 2          # pgrm.py:547-548: prior_fusion runs iff the prior has 2 channels, whatever `mode` is
    if fuse and m.mode:
        raise _abi.DpmnError("PGRM(mode=True) has no prior_fusion: x_q must have 3 channels (pgrm.py:470,547)")
    if NATIVE_FWD and FUSED_ATTN and FUSED_ATTN_BWD and FUSED_SKMLP and x_q.dtype == torch.float32 and x_kv.dtype == torch.float32:
        import ctypes as C
        w = m._weights()
        if lib.dpmn_pgrm_forward_train_supported(C.byref(w), B):
            return _forward_native(m, x_q, x_kv, residuals, drop, w)
    pf = (m.prior_fusion.weight, m.prior_fusion.bias) if fuse else (None, None)
    pd = drop["p"] if drop else 0.0
    pa = drop["pa"] if drop else 0.0
    sd = drop["seeds"] if drop else [0] * N_SEEDS
    # pos_drop (pgrm.py:550-551) in the patch embedding's epilogue
    tq = ops.patch_embed_ln(x_q, pe.proj.weight, pe.proj.bias, pe.norm.weight, pe.norm.bias, m.patch, *pf, p_drop=pd, seed=sd[0]).reshape(M, Cd)
    tkv = ops.patch_embed_ln(x_kv, pe.proj.weight, pe.proj.bias, pe.norm.weight, pe.norm.bias, m.patch, p_drop=pd, seed=sd[1]).reshape(M, Cd)
    sv = dict(x_q=x_q, x_kv=x_kv, tq=tq, residuals=list(residuals), blocks=[], drop=drop)
    parts = (L + 31) // 32
    for bi, blk in enumerate(m.layers[0].blocks):
        a, sk, mlp = blk.attn, blk.attn.sknet, blk.mlp
        win = [min(H, Wd) if min(H, Wd) <= w else w for w in m.window_size]
        shift = [0 if (bi == 0 or min(H, Wd) <= w) else w // 2 for w in m.window_size]
        tables = [getattr(a, "relative_position_bias_table_%d" % g) for g in range(G)]
        s = dict(tkv_in=tkv, win=win, shift=shift, tables=tables)
        dpb = drop["dp"][bi] if drop else 0.0
        sb = sd[2 + 5 * bi:7 + 5 * bi]
        if FUSED_ATTN and ops.ln_qkv_window_attn_supported(Cd, win, hpg, H, Wd):
            # norm1_q / norm1_kv + q / kv Linear + the three window sizes (+ attn_drop) in one launch; q and kv come back for the
            # backward, the normalised tokens are NOT kept (the backward recomputes them: two LayerNorm launches there instead of
            # two LayerNorm + two GEMM + three attention launches and 2 x 19 MB of saved activations here)
            # (FUSED_ATTN_BWD: q / kv are not even written -- the backward kernel repeats the projection from tq / tkv and the folded
            #  weights of this call, kept in s["fold"])
            fold = []
            cat, q_, kv_ = ops.ln_qkv_window_attn_train(tq.reshape(B, L, Cd), tkv.reshape(B, L, Cd), blk.norm1_q.weight, blk.norm1_q.bias,
                                                        blk.norm1_kv.weight, blk.norm1_kv.bias, a.q.weight, a.q.bias, a.kv.weight, a.kv.bias,
                                                        tables, win, shift, hpg, H, Wd, p_drop=pa, seed=sb[0], save_qkv=not FUSED_ATTN_BWD,
                                                        fold_out=fold)
            s["cat"], s["fold"] = cat.reshape(M, Cd), fold[0]
            s["q"], s["kv"] = (None, None) if FUSED_ATTN_BWD else (q_.reshape(M, Cd), kv_.reshape(M, 2 * Cd))
            s["nq"] = s["nkv"] = None
        else:
            s["nq"] = layernorm(tq, blk.norm1_q.weight, blk.norm1_q.bias)
            s["nkv"] = layernorm(tkv, blk.norm1_kv.weight, blk.norm1_kv.bias)
            s["q"] = ops.linear(s["nq"], a.q.weight, a.q.bias)
            s["kv"] = ops.linear(s["nkv"], a.kv.weight, a.kv.bias)
            s["cat"] = ops.window_attn(s["q"].reshape(B, L, Cd), s["kv"].reshape(B, L, 2 * Cd), tables, win, shift, hpg, H, Wd,
                                       p_drop=pa, seed=sb[0]).reshape(M, Cd)
        s["feats"] = _e(M, Cd, like=tkv)
        s["partial"] = _e(B * parts, Cd, like=tkv)
        check(lib.dpmn_sk_proj_f32(dptr(s["cat"]), dptr(sk.proj.weight), dptr(sk.proj.bias), dptr(s["feats"]), dptr(s["partial"]), M, Cd, stream()))
        s["avec"] = _e(B, G, Cd // G, like=tkv)
        check(lib.dpmn_sk_gate_f32(dptr(s["partial"]), parts, L, dptr(sk.fc1.weight), dptr(sk.fc1.bias), dptr(sk.fc2.weight),
                                   dptr(sk.fc2.bias), dptr(s["avec"]), B, Cd, G, sk.fc1.weight.shape[0], stream()))
        if FUSED_SKMLP and ops.sk_mlp_in_supported(M, L, Cd, G, Ch):
            # select + proj_head + both residuals (DropPath on the attention branch included) -> x1 -> LayerNorm2 -> fc1 in one launch
            # (gemm.hip k_sk_mlp_in), V and n2 written on the way for the backward's weight-gradient GEMMs: replaces four / five launches
            s["x1"], s["ypre"], s["V"], s["n2"] = ops.sk_mlp_in(s["cat"], s["avec"], sk.proj_head.weight, sk.proj_head.bias, s["feats"], tkv,
                                                                 blk.norm2.weight, blk.norm2.bias, mlp.fc1.weight, mlp.fc1.bias, L, save=True,
                                                                 p_row=dpb, seed_row=sb[1])
        else:
            s["V"] = _e(M, Cd // G, like=tkv)
            check(lib.dpmn_sk_select_only_f32(dptr(s["cat"]), dptr(s["avec"]), dptr(s["V"]), M, L, Cd, G, stream()))
            if dpb > 0:      # x1 = shortcut + DropPath(attention branch)
                branch = ops.linear(s["V"], sk.proj_head.weight, sk.proj_head.bias, res1=s["feats"])
                s["x1"] = ops.dropout(branch, p_row=dpb, seed_row=sb[1], row_len=L * Cd, res=tkv)
            else:
                s["x1"] = ops.linear(s["V"], sk.proj_head.weight, sk.proj_head.bias, res1=s["feats"], res2=tkv)
            s["n2"] = layernorm(s["x1"], blk.norm2.weight, blk.norm2.bias)
            s["ypre"] = ops.linear(s["n2"], mlp.fc1.weight, mlp.fc1.bias)
        # the two GELUs and the element dropout between fc1's GELU and the conv ride in the depthwise conv (on load / second output)
        r = int(round(L ** 0.5))
        s["gpre"], s["g"] = _e(M, Ch, like=tkv), _e(M, Ch, like=tkv)
        check(lib.dpmn_dwconv3x3_train_f32(dptr(s["ypre"]), dptr(mlp.depthwise_conv.weight), dptr(mlp.depthwise_conv.bias),
                                           dptr(s["gpre"]), dptr(s["g"]), 1, float(pd), int(sb[2]), B, Ch, r, stream()))
        s["z"] = ops.pointwise(s["g"].reshape(B, L, Ch), mlp.pointwise_conv.weight.reshape(Ch, Ch), mlp.pointwise_conv.bias).reshape(M, Ch)
        if (pd > 0 or dpb > 0) and M % 64 == 0 and Cd % 96 == 0 and Ch % 32 == 0 and Ch > 192:
            # x_kv = x1 + DropPath(Dropout(fc2(z))): both masks in the GEMM's epilogue
            tkv = torch.empty(M, Cd, device=tkv.device)
            check(lib.dpmn_linear_drop_f32(dptr(s["z"]), dptr(mlp.fc2.weight), dptr(mlp.fc2.bias), dptr(s["x1"]), dptr(tkv), M, Cd, Ch, float(pd),
                                           int(sb[3]), float(dpb), int(sb[4]), L * Cd, stream()))
        elif pd > 0 or dpb > 0:
            branch = ops.linear(s["z"], mlp.fc2.weight, mlp.fc2.bias)
            tkv = ops.dropout(branch, pd, sb[3], dpb, sb[4], row_len=L * Cd, res=s["x1"])
        else:
            tkv = ops.linear(s["z"], mlp.fc2.weight, mlp.fc2.bias, res1=s["x1"])
        sv["blocks"].append(s)
    sv["tkv_out"] = tkv
    c0s, c1s = ConvSpec(m.conv_before_upsample[0].weight, m.conv_before_upsample[0].bias), ConvSpec(m.conv_before_upsample[1].weight, m.conv_before_upsample[1].bias)
    sv["c0"] = c0s.forward(tkv.reshape(B, H, Wd, Cd))
    sv["c1"] = c1s.forward(sv["c0"])
    wl = [getattr(m, "weight_list_%d" % i) for i in range(m.iter + 1)]
    out = torch.empty(B, m.hidden_size, m.img_size[0], m.img_size[1], device=x_kv.device)
    check(lib.dpmn_pgrm_tail_elem_f32(dptr(sv["c1"]), _abi.ptr_array(wl), _abi.ptr_array(residuals), len(residuals), dptr(out), B, H, Wd, stream()))
    return out, sv


PE_WGRAD_IN_KERNEL = os.environ.get("DPMN_PE_WGRAD_IN_KERNEL", "1") != "0"      # 0: PatchEmbed weight gradient as dconv^T . patches (gemm_tn)
NATIVE_BWD = os.environ.get("DPMN_PGRM_NATIVE_BWD", "1") != "0"      # 0: the Swin-block loop of the backward op by op from Python


def _native_blocks_ok(m, sv, B):
    """The two-block loop of backward() as one native call (csrc/pgrm_backward.hip)?  Needs the fused training geometry, the
    atomics-free forms (the only ones the driver issues), deferred reductions (their arena is where its partial rows go) and the
    recomputing attention backward (q / kv not saved)."""
    if not (NATIVE_BWD and DET_SMALL and LNB_DET and FUSED_ATTN_BWD and _tn_pending is not None):
        return False
    if any(s.get("q") is not None or s.get("fold") is None for s in sv["blocks"]) or not sv["tq"].is_cuda:
        return False
    import ctypes as C
    return bool(lib.dpmn_pgrm_forward_train_supported(C.byref(m._weights()), B))


def _blocks_backward_native(m, sv, gr, drop, dtkv, dtq, dcat, B):
    import ctypes as C
    w = m._weights()
    G = len(m.window_size)
    blocks = m.layers[0].blocks
    # gradient sinks in the layout of the weight table; in direct mode (flat gradient arena) they never move: built once per module
    cached = m.__dict__.get("_bwd_sinks")
    key = tuple(gr[p].data_ptr() for p in (blocks[0].mlp.fc2.weight, blocks[1].attn.q.bias))
    if cached is None or cached[0] != key:
        gs = (_abi.PgrmBlock * 2)()
        for bi, blk in enumerate(blocks):
            a, sk, mlp, g = blk.attn, blk.attn.sknet, blk.mlp, gs[bi]
            d = lambda p: dptr(gr[p])
            g.norm1_q_w, g.norm1_q_b, g.norm1_kv_w, g.norm1_kv_b = d(blk.norm1_q.weight), d(blk.norm1_q.bias), d(blk.norm1_kv.weight), d(blk.norm1_kv.bias)
            g.q_w, g.q_b, g.kv_w, g.kv_b = d(a.q.weight), d(a.q.bias), d(a.kv.weight), d(a.kv.bias)
            for k in range(G):
                g.bias_table[k] = d(getattr(a, "relative_position_bias_table_%d" % k))
            g.sk_proj_w, g.sk_proj_b = d(sk.proj.weight), d(sk.proj.bias)
            g.sk_fc1_w, g.sk_fc1_b, g.sk_fc2_w, g.sk_fc2_b = d(sk.fc1.weight), d(sk.fc1.bias), d(sk.fc2.weight), d(sk.fc2.bias)
            g.sk_head_w, g.sk_head_b = d(sk.proj_head.weight), d(sk.proj_head.bias)
            g.norm2_w, g.norm2_b = d(blk.norm2.weight), d(blk.norm2.bias)
            g.fc1_w, g.fc1_b, g.fc2_w, g.fc2_b = d(mlp.fc1.weight), d(mlp.fc1.bias), d(mlp.fc2.weight), d(mlp.fc2.bias)
            g.dw_w, g.dw_b = d(mlp.depthwise_conv.weight), d(mlp.depthwise_conv.bias)
            g.pw_w, g.pw_b = d(mlp.pointwise_conv.weight), d(mlp.pointwise_conv.bias)
        cached = (key, gs)
        if getattr(m, "_dpmn_bucket", None) is not None:
            m.__dict__["_bwd_sinks"] = cached
    gs = cached[1]
    # transposed weights of the data-gradient products (the step's pack cache refreshes them in its one launch)
    ts, keep = (_abi.PgrmBlockT * 2)(), []
    Ch = int(m.embed_dim * m.mlp_ratio)
    for bi, blk in enumerate(blocks):
        a, sk, mlp = blk.attn, blk.attn.sknet, blk.mlp
        for name, wgt in (("fc2_t", mlp.fc2.weight), ("pw_t", mlp.pointwise_conv.weight.reshape(Ch, Ch)), ("fc1_t", mlp.fc1.weight),
                          ("head_t", sk.proj_head.weight), ("proj_t", sk.proj.weight), ("q_t", a.q.weight), ("kv_t", a.kv.weight)):
            t_ = packing.transposed(wgt)
            keep.append(t_)
            setattr(ts[bi], name, dptr(t_))
    cs = sv.get("_cs")
    if cs is None:      # the per-op forward's dictionary: the same tensors, as the struct
        cs = _abi.PgrmSaved()
        cs.tq, cs.tkv0 = dptr(sv["tq"]), dptr(sv["blocks"][0]["tkv_in"])
        for bi in range(2):
            s = sv["blocks"][bi]
            for name in _abi.PgrmSavedBlock.NAMES:
                if name == "tkv_out":
                    t_ = sv["blocks"][1]["tkv_in"] if bi == 0 else sv["tkv_out"]
                else:
                    t_ = s[name]
                setattr(cs.blk[bi], name, dptr(t_))
    cd = None
    if drop:
        cd = _abi.PgrmDrop()
        cd.p, cd.pa = drop["p"], drop["pa"]
        cd.dp[0], cd.dp[1] = drop["dp"]
        for i, v in enumerate(drop["seeds"]):
            cd.seeds[i] = v
    numel = _abi.int_array([t.numel() for t in sv["blocks"][0]["tables"]])
    need = lib.dpmn_pgrm_blocks_backward_scratch_bytes(C.byref(w), B, numel)
    scratch = torch.empty(need // 4, device=dtkv.device)
    arena = _tn_pending[0]
    used = C.c_size_t(_tn_pending[1])
    _tn_pending[2].append(scratch)       # holds partial rows of queued reductions: alive until the flush
    check(lib.dpmn_pgrm_blocks_backward_leaf_f32(C.byref(w), gs, ts, C.byref(cs), C.byref(cd) if cd is not None else None, numel, dptr(dtkv), dptr(dtq),
                                                 _abi.ptr_array(dcat), dptr(_zero_bias(Ch, dtkv.device)), scratch.data_ptr(), need, arena.data_ptr(),
                                                 arena.numel() * 4, C.byref(used), B, stream(), _leaf_stream(dtkv.device)))
    _tn_pending[1] = used.value
    del keep


# 1: the weight gradients of the Swin blocks (leaves of the backward graph) go to a second stream (csrc/pgrm_backward.hip).  Which one: a
# stream that is idle while the PGRM backwards run AND sits on another hardware queue than the calling branch stream (HIP maps streams
# onto four queues in creation order, dpmn_amd/_streams.py) -- the conv weight-gradient stream of the CMM backward for branch 1 (and for
# an unforked step), the PSN prefetch lane for branch 2.
BWD_LEAF_STREAM = os.environ.get("DPMN_BWD_LEAF_STREAM", "0") != "0"


def _leaf_stream(device):
    if not BWD_LEAF_STREAM or torch.cuda.is_current_stream_capturing():
        return None
    from .. import _streams
    p = _streams.pool(device)
    cur = torch.cuda.current_stream(device)
    leaf = p["psn"] if cur == p["branch"][1] else p["wgrad"]
    return None if leaf == cur else leaf.cuda_stream


def backward(m, sv, dout, need_dx_kv=True):
    """Returns (dx_kv or None, [dresidual_i or None], {param: grad}, direct) -- see grad_targets."""
    B = sv["x_kv"].shape[0]
    H, Wd = m.patches_resolution
    L, Cd = H * Wd, m.embed_dim
    M, Ch, G = B * L, int(m.embed_dim * m.mlp_ratio), len(m.window_size)
    hpg = m.num_heads // G
    parts = (L + 31) // 32
    gr, direct = grad_targets(m)
    residuals = sv["residuals"]
    # every zero-initialised accumulator of this call comes out of ONE zero-filled slab (one memset instead of ~12 fills)
    img = sv["x_kv"]
    sizes = [r.numel() for r in residuals[1:]] + [M * Cd] + ([] if DET_SMALL else [M * Cd, B * Cd] * 2) + [Cd * 16, 16 * Cd] * 2 + [img.numel()]
    slab = torch.zeros(sum((n + 63) // 64 * 64 for n in sizes), device=dout.device)
    _off = [0]

    def zeros(*shape):
        n = 1
        for d_ in shape:
            n *= d_
        v = slab[_off[0]:_off[0] + n].view(*shape)
        _off[0] += (n + 63) // 64 * 64
        return v
    dres = [None] + [zeros(*r.shape) for r in residuals[1:]]
    wl = [getattr(m, "weight_list_%d" % i) for i in range(m.iter + 1)]
    dwl = [gr[w] for w in wl]
    dout = dout.contiguous()
    dc1 = torch.empty_like(sv["c1"])
    n_res = len(residuals)
    dres_ptrs = (_abi.fp * max(n_res, 1))()
    for i in range(1, n_res):
        dres_ptrs[i] = dptr(dres[i])
    check(lib.dpmn_pgrm_tail_elem_bwd_f32(dptr(dout), dptr(sv["c1"]), _abi.ptr_array(wl), _abi.ptr_array(residuals), dres_ptrs,
                                          _abi.ptr_array(dwl), n_res, dptr(dc1), B, H, Wd, stream()))
    c0m, c1m = m.conv_before_upsample[0], m.conv_before_upsample[1]
    dc0 = ConvSpec(c1m.weight, c1m.bias).backward(sv["c0"], dc1, gr[c1m.weight], gr[c1m.bias])
    dtkv = ConvSpec(c0m.weight, c0m.bias).backward(sv["tkv_out"].reshape(B, H, Wd, Cd), dc0, gr[c0m.weight], gr[c0m.bias]).reshape(M, Cd)
    dtq = zeros(M, Cd)
    drop = sv.get("drop")
    pd = drop["p"] if drop else 0.0
    pa = drop["pa"] if drop else 0.0
    sd = drop["seeds"] if drop else [0] * N_SEEDS
    native = _native_blocks_ok(m, sv, B)
    if native:
        _blocks_backward_native(m, sv, gr, drop, dtkv, dtq, [torch.empty(M, Cd, device=dout.device) for _ in range(2)], B)      # dtkv updated in place
    for bi in (() if native else (1, 0)):
        blk = m.layers[0].blocks[bi]
        a, sk, mlp = blk.attn, blk.attn.sknet, blk.mlp
        s = sv["blocks"][bi]
        dpb = drop["dp"][bi] if drop else 0.0
        sb = sd[2 + 5 * bi:7 + 5 * bi]
        dx2 = dtkv
        # fc2 (+ residual x1); with dropout the branch gradient is dx2 under the same masks
        dbr = ops.dropout(dx2, pd, sb[3], dpb, sb[4], row_len=L * Cd, out=torch.empty_like(dx2)) if (pd > 0 or dpb > 0) else dx2
        dz = linear_bwd(dbr, s["z"], mlp.fc2.weight, gr[mlp.fc2.weight], gr[mlp.fc2.bias])
        # pointwise conv on the raw (B, Ch, L) views
        wp = mlp.pointwise_conv.weight.reshape(Ch, Ch)
        dg = ops.pointwise(dz.reshape(B, L, Ch), packing.transposed(wp), _zero_bias(Ch, dz.device)).reshape(M, Ch)
        # partial rows of the atomics-free pointwise / bias / depthwise-conv gradients: slices of the backward's arena
        if DET_SMALL:
            ptr, nb = _ws(dz.device, lib.dpmn_pointwise_wgrad_det_bytes(Ch, L))
            check(lib.dpmn_pointwise_wgrad_det_f32(dptr(dz), dptr(s["g"]), dptr(gr[mlp.pointwise_conv.weight]), B, Ch, L, ptr, nb, stream()))
            _ws_done()
        else:
            check(lib.dpmn_pointwise_wgrad_f32(dptr(dz), dptr(s["g"]), dptr(gr[mlp.pointwise_conv.weight]), B, Ch, L, stream()))
        if DET_SMALL:
            ptr, nb = _ws(dz.device, B * Ch * 4)
            check(lib.dpmn_rowsum_mod_det_f32(dptr(dz), dptr(gr[mlp.pointwise_conv.bias]), B * Ch, L, Ch, ptr, nb, stream()))
            _ws_done()
        else:
            check(lib.dpmn_rowsum_mod_f32(dptr(dz), dptr(gr[mlp.pointwise_conv.bias]), B * Ch, L, Ch, stream()))
        # GELU'(gpre) on the way in, GELU (+ the dropout mask) on the forward input, mask and GELU'(ypre) on the way out
        dypre = torch.empty_like(dg)
        r = int(round(L ** 0.5))
        if DET_SMALL:
            ptr, nb = _ws(dz.device, lib.dpmn_dwconv3x3_bwd_det_bytes(B, Ch, r))
            check(lib.dpmn_dwconv3x3_bwd_fused_det_f32(dptr(s["ypre"]), dptr(dg), dptr(s["gpre"]), dptr(mlp.depthwise_conv.weight),
                                                       dptr(dypre), dptr(gr[mlp.depthwise_conv.weight]), dptr(gr[mlp.depthwise_conv.bias]),
                                                       1, 1, float(pd), int(sb[2]), B, Ch, r, ptr, nb, stream()))
            _ws_done()
        else:
            check(lib.dpmn_dwconv3x3_bwd_fused_f32(dptr(s["ypre"]), dptr(dg), dptr(s["gpre"]), dptr(mlp.depthwise_conv.weight),
                                                   dptr(dypre), dptr(gr[mlp.depthwise_conv.weight]), dptr(gr[mlp.depthwise_conv.bias]),
                                                   1, 1, float(pd), int(sb[2]), B, Ch, r, stream()))
        dn2 = linear_bwd(dypre, s["n2"], mlp.fc1.weight, gr[mlp.fc1.weight], gr[mlp.fc1.bias])
        dx1 = dx2      # in place: every reader of dx2 (fc2's two backward GEMMs, the dropout copy) is already queued on this stream
        layernorm_bwd(s["x1"], dn2, blk.norm2.weight, dx1, True, gr[blk.norm2.weight], gr[blk.norm2.bias])
        # x1 = tkv_in + DropPath(feats + V Wh^T + bh)
        dat = ops.dropout(dx1, p_row=dpb, seed_row=sb[1], row_len=L * Cd, out=torch.empty_like(dx1)) if dpb > 0 else dx1
        dV = linear_bwd(dat, s["V"], sk.proj_head.weight, gr[sk.proj_head.weight], gr[sk.proj_head.bias])
        dcat = torch.empty(M, Cd, device=dout.device) if DET_SMALL else zeros(M, Cd)      # DET_SMALL: written by the select backward
        dS = torch.empty(B, Cd, device=dout.device)
        dmid = sk.fc1.weight.shape[0]
        if DET_SMALL:
            # the gate's gradient dA as per-block partial rows added in block order by the gate backward (it feeds the DATA path:
            # with atomics every gradient upstream of this block differed from run to run), the gate's weight gradients as
            # per-image rows added in image order by the backward's one reduce launch
            dA = torch.empty(parts, B, Cd, device=dout.device)
            check(lib.dpmn_sk_select_bwd_det_set_f32(dptr(s["cat"]), dptr(s["avec"]), dptr(dV), dptr(dcat), dptr(dA), B, L, Cd, G, stream()))
            wp2, wp1 = torch.empty(B, Cd * dmid + Cd, device=dout.device), torch.empty(B, dmid * Cd + dmid, device=dout.device)
            check(lib.dpmn_sk_gate_bwd_det_f32(dptr(s["partial"]), parts, L, dptr(sk.fc1.weight), dptr(sk.fc1.bias), dptr(sk.fc2.weight),
                                               dptr(s["avec"]), dptr(dA), parts, dptr(dS), dptr(wp2), dptr(wp1), B, Cd, G, dmid, stream()))
            defer_rows(wp2, gr[sk.fc2.weight], gr[sk.fc2.bias], Cd * dmid, Cd, B)
            defer_rows(wp1, gr[sk.fc1.weight], gr[sk.fc1.bias], dmid * Cd, dmid, B)
        else:
            dA = zeros(B, G, Cd // G)
            check(lib.dpmn_sk_select_bwd_f32(dptr(s["cat"]), dptr(s["avec"]), dptr(dV), dptr(dcat), dptr(dA), B, L, Cd, G, stream()))
            check(lib.dpmn_sk_gate_bwd_f32(dptr(s["partial"]), parts, L, dptr(sk.fc1.weight), dptr(sk.fc1.bias), dptr(sk.fc2.weight),
                                           dptr(s["avec"]), dptr(dA), dptr(dS), dptr(gr[sk.fc1.weight]), dptr(gr[sk.fc1.bias]),
                                           dptr(gr[sk.fc2.weight]), dptr(gr[sk.fc2.bias]), B, Cd, G, dmid, stream()))
        dfeats = torch.empty(M, Cd, device=dout.device)
        check(lib.dpmn_sk_feats_grad_f32(dptr(dat), dptr(s["feats"]), dptr(dS), dptr(dfeats), M, L, Cd, stream()))
        gemm_tn(dfeats, s["cat"], gr[sk.proj.weight], gr[sk.proj.bias])
        dcat = ops.linear(dfeats, packing.transposed(sk.proj.weight), None, res1=dcat)
        # window attention
        dtab = [gr[t] for t in s["tables"]]
        if s["q"] is None:
            # one launch: q / k / v recomputed, all three window sizes on MFMA, bias-table gradients as per-block partial rows
            dq, dkv, tparts = ops.ln_qkv_window_attn_bwd(sv["tq"].reshape(B, L, Cd), s["tkv_in"].reshape(B, L, Cd), blk.norm1_q.weight,
                                                         blk.norm1_q.bias, blk.norm1_kv.weight, blk.norm1_kv.bias, a.q.weight, a.q.bias,
                                                         a.kv.weight, a.kv.bias, s["tables"], s["win"], s["shift"], hpg, H, Wd,
                                                         dcat.reshape(B, L, Cd), p_drop=pa, seed=sb[0], fold=s["fold"])
            for g_ in range(G):
                defer_rows(tparts[g_], dtab[g_], None, s["tables"][g_].numel(), 0, tparts[g_].shape[0])
        else:
            dq = torch.empty(M, Cd, device=dout.device)
            dkv = torch.empty(M, 2 * Cd, device=dout.device)
        if s["q"] is not None and DET_SMALL:
            # bias-table gradients as per-block partial rows, added in block order by the backward's one reduce launch
            nrows = lib.dpmn_window_attn_bwd_part_rows(B, H, Wd)
            tparts = [torch.empty(nrows, t.numel(), device=dout.device) for t in s["tables"]]
            rows = _abi.int_array([0] * G)
            check(lib.dpmn_window_attn_drop_bwd_det_f32(dptr(s["q"]), dptr(s["kv"]), _abi.ptr_array(s["tables"]), _abi.int_array(s["win"]),
                                                        _abi.int_array(s["shift"]), G, hpg, dptr(dcat), dptr(dq), dptr(dkv),
                                                        _abi.ptr_array(tparts), rows, B, H, Wd, Cd, float(pa), int(sb[0]), stream()))
            for g_ in range(G):
                defer_rows(tparts[g_], dtab[g_], None, s["tables"][g_].numel(), 0, rows[g_])
        elif s["q"] is not None:
            check(lib.dpmn_window_attn_drop_bwd_f32(dptr(s["q"]), dptr(s["kv"]), _abi.ptr_array(s["tables"]), _abi.int_array(s["win"]),
                                                    _abi.int_array(s["shift"]), G, hpg, dptr(dcat), dptr(dq), dptr(dkv),
                                                    _abi.ptr_array(dtab), B, H, Wd, Cd, float(pa), int(sb[0]), stream()))
        nq = s["nq"] if s["nq"] is not None else layernorm(sv["tq"], blk.norm1_q.weight, blk.norm1_q.bias)
        dnq = linear_bwd(dq, nq, a.q.weight, gr[a.q.weight], gr[a.q.bias])
        del nq
        layernorm_bwd(sv["tq"], dnq, blk.norm1_q.weight, dtq, True, gr[blk.norm1_q.weight], gr[blk.norm1_q.bias])
        nkv = s["nkv"] if s["nkv"] is not None else layernorm(s["tkv_in"], blk.norm1_kv.weight, blk.norm1_kv.bias)
        dnkv = linear_bwd(dkv, nkv, a.kv.weight, gr[a.kv.weight], gr[a.kv.bias])
        del nkv
        layernorm_bwd(s["tkv_in"], dnkv, blk.norm1_kv.weight, dx1, True, gr[blk.norm1_kv.weight], gr[blk.norm1_kv.bias])
        dtkv = dx1
    if pd > 0 and not DET_SMALL:       # pos_drop on both token streams (pgrm.py:554-555); DET_SMALL: applied by the patch-embed backward on load
        ops.dropout(dtq, pd, sd[0])
        ops.dropout(dtkv, pd, sd[1])
    # patch embeddings (shared weights): kv path gives the image gradient, q path the prior_fusion gradients
    pe = m.patch_embed
    dx_kv = None
    for which, img, dtok in (("kv", sv["x_kv"], dtkv), ("q", sv["x_q"], dtq)):
        fuse = which == "q" and img.shape[1] == 2
        pfw, pfb = (m.prior_fusion.weight, m.prior_fusion.bias) if fuse else (None, None)
        dconv = torch.empty(M, Cd, device=dout.device)
        patches = None if (DET_SMALL and PE_WGRAD_IN_KERNEL and Cd == 96) else torch.empty(M, 16, device=dout.device)
        pe_wg = DET_SMALL and PE_WGRAD_IN_KERNEL and Cd == 96
        if pe_wg:
            # as the branch below, with the conv's weight / bias gradient as per-block partial rows out of the same kernel (no `patches`
            # tensor, no skinny dconv^T . patches GEMM with its own reduce launch and the torch add behind it): one (2 nr, 13 C) buffer
            # and one pending sum for both token streams
            nr = (M + 63) // 64
            if which == "kv":
                lnp = torch.empty(2 * nr, 2 * Cd, device=dout.device)
                pew = torch.empty(2 * nr, 13 * Cd, device=dout.device)
            off = 0 if which == "kv" else nr
            # the image gradient of the kv stream straight out of the kernel (3 input channels, no prior_fusion)
            direct_dx = which == "kv" and need_dx_kv and img.shape[1] == 3
            if direct_dx:
                dx_kv = torch.empty_like(img)
            check(lib.dpmn_patch_embed_bwd_det_wgrad_f32(dptr(img), img.shape[1], dptr(pfw, True), dptr(pfb, True), dptr(pe.proj.weight),
                                                         dptr(pe.proj.bias), dptr(pe.norm.weight), dptr(dtok), dptr(dconv),
                                                         lnp.data_ptr() + off * 2 * Cd * 4, pew.data_ptr() + off * 13 * Cd * 4,
                                                         dptr(dx_kv) if direct_dx else None, B, img.shape[2], img.shape[3], Cd, float(pd),
                                                         int(sd[1] if which == "kv" else sd[0]), stream()))
            if which == "q":
                defer_rows(lnp, gr[pe.norm.weight], gr[pe.norm.bias], Cd, Cd, 2 * nr)
                defer_rows(pew, gr[pe.proj.weight], gr[pe.proj.bias], 12 * Cd, Cd, 2 * nr)
        elif DET_SMALL:      # LayerNorm parameter gradients as per-block partial rows, added in block order at the end of the backward
            # (ONE buffer and one pending sum for both token streams: the patch embedding is shared, and two pending sums into the
            #  same tensor would race in the multi-descriptor reduce launch)
            nr = (M + 63) // 64
            if which == "kv":
                lnp = torch.empty(2 * nr, 2 * Cd, device=dout.device)
            check(lib.dpmn_patch_embed_bwd_det_drop_f32(dptr(img), img.shape[1], dptr(pfw, True), dptr(pfb, True), dptr(pe.proj.weight),
                                                        dptr(pe.proj.bias), dptr(pe.norm.weight), dptr(dtok), dptr(dconv), dptr(patches),
                                                        lnp.data_ptr() + (0 if which == "kv" else nr * 2 * Cd * 4), B, img.shape[2], img.shape[3], Cd,
                                                        float(pd), int(sd[1] if which == "kv" else sd[0]), stream()))
            if which == "q":
                defer_rows(lnp, gr[pe.norm.weight], gr[pe.norm.bias], Cd, Cd, 2 * nr)
        else:
            check(lib.dpmn_patch_embed_bwd_f32(dptr(img), img.shape[1], dptr(pfw, True), dptr(pfb, True), dptr(pe.proj.weight),
                                               dptr(pe.proj.bias), dptr(pe.norm.weight), dptr(dtok), dptr(dconv), dptr(patches),
                                               dptr(gr[pe.norm.weight]), dptr(gr[pe.norm.bias]), B, img.shape[2], img.shape[3], Cd, stream()))
        if not pe_wg:
            dw16 = zeros(Cd, 16)
            gemm_tn(dconv, patches, dw16, gr[pe.proj.bias], leaf=False)      # dw16 is read right below
            gr[pe.proj.weight] += dw16[:, :12].reshape(pe.proj.weight.shape)
        need_din = fuse or (which == "kv" and need_dx_kv)
        if need_din and not (pe_wg and which == "kv" and img.shape[1] == 3):
            w16 = zeros(16, Cd)
            w16[:12] = pe.proj.weight.reshape(Cd, 12).t()
            din = ops.linear(dconv, w16)
            if fuse and DET_SMALL:
                pfp = torch.empty((B * img.shape[2] * img.shape[3] + 255) // 256, 57, device=dout.device)
                check(lib.dpmn_prior_fusion_wgrad_det_f32(dptr(din), dptr(img), dptr(pfp), B, img.shape[2], img.shape[3], stream()))
                defer_rows(pfp, gr[m.prior_fusion.weight], gr[m.prior_fusion.bias], 54, 3, pfp.shape[0])
            elif fuse:
                check(lib.dpmn_prior_fusion_wgrad_f32(dptr(din), dptr(img), dptr(gr[m.prior_fusion.weight]), dptr(gr[m.prior_fusion.bias]),
                                                      B, img.shape[2], img.shape[3], stream()))
            else:
                dx_kv = zeros(*img.shape)
                check(lib.dpmn_patch_scatter_f32(dptr(din), dptr(dx_kv), img.shape[1], B, img.shape[2], img.shape[3], stream()))
    return dx_kv, dres, gr, direct


def backward_deferred(m, sv, dout, need_dx_kv=True):
    """backward() with the ordered reductions of the call queued into ONE multi-descriptor launch at its end (tn_flush)."""
    global _tn_pending
    if TN_DEFER and not torch.cuda.is_current_stream_capturing():
        _tn_pending = [_tn_workspace(dout.device), 0, []]
        check(lib.dpmn_reduce_defer_begin())
    try:
        res = backward(m, sv, dout, need_dx_kv=need_dx_kv)
        tn_flush(end=True)          # every parameter gradient is in place before the bucket is signalled
    finally:
        if _tn_pending is not None:
            lib.dpmn_reduce_defer_flush(1, stream())      # (after an exception: drop the queue, leave deferral off)
        _tn_pending = None
    return res


class PGRMFunction(torch.autograd.Function):
    """autograd bridge: inputs (x_q, x_kv, n_res, *residuals, *params)."""

    @staticmethod
    def forward(ctx, m, x_q, x_kv, n_res, *rest):
        residuals = list(rest[:n_res])
        out, sv = forward(m, x_q.contiguous().float(), x_kv.contiguous().float(), [r.contiguous().float() for r in residuals],
                          drop=drop_config(m))
        ctx.m, ctx.sv, ctx.n_res = m, sv, n_res
        ctx.need_kv = x_kv.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.m
        dx_kv, dres, gr, direct = backward_deferred(m, ctx.sv, dout, ctx.need_kv)
        ctx.sv = None
        dres_out = [None if d is None else d for d in dres] + [None] * (ctx.n_res - len(dres))
        return (None, None, dx_kv, None) + tuple(dres_out[:ctx.n_res]) + finish_grads(m, gr, direct)


def apply(m, x_q, x_kv, residual_list):
    if getattr(m, "_dpmn_bucket", None) is not None and torch.is_grad_enabled():
        m._dpmn_bucket.note_use()       # a shared module (--sr_share) reports ready after as many backward calls
    return PGRMFunction.apply(m, x_q, x_kv, len(residual_list), *residual_list, *fn_inputs(m))

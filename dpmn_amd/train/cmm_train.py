"""Training-mode CMM (model/cmm.py:120-161): train-mode BatchNorm (batch statistics, cmm.py:12) forward and the full
explicit backward, composed from the NHWC implicit-GEMM conv kernels.

Every conv writes its RAW output plus per-channel sum / sum-of-squares (conv epilogue); BatchNorm becomes a per-channel
(scale, shift) that the CONSUMERS apply on load together with their LeakyReLU / ReLU -- normalised tensors are never
materialised.  Backward walks the fixed graph in reverse: consumer data-gradient -> activation backward through the
affine (accumulated per producer) -> BatchNorm backward through the batch statistics -> producer weight / data gradients.
"""
import ctypes as C

import os

import torch
from .._abi import stream_of as _abi_stream_of

from .. import ops
from .._abi import dptr, lib, check, stream
from ..model import packing
from .pgrm_train import colsum, conv_wgrad_into, grad_targets, finish_grads, params_of, fn_inputs

ACT = ops.ACT


class T:
    """raw NHWC tensor + optional BatchNorm affine; consumers see act(scale * r + shift)."""

    def __init__(self, r, bn=None, scale=None, shift=None, mean=None, rstd=None):
        self.r, self.bn, self.scale, self.shift, self.mean, self.rstd = r, bn, scale, shift, mean, rstd
        self.G = None     # dL/d(scale*r+shift), accumulated over consumers

    @property
    def aff(self):
        return None if self.scale is None else (self.scale, self.shift)

    def grad_buf(self):
        if self.G is None:
            self.G = torch.zeros_like(self.r)
        return self.G


_STATS = {}


def stats_buffer(cout, device):
    """(32, 2, cout) zeroed fp64 BatchNorm-statistics slots for ONE convolution (order-independent sums: conv.hip STAT_SLOTS).  A single persistent buffer per device: the conv that
    fills it and the _bn_finalize that reads it are stream-ordered, and the finalize kernel zeroes what it read, so the next
    convolution gets the same memory back clean (33 memsets per step otherwise)."""
    key = (device, _abi_stream_of(device))
    buf = _STATS.get(key)
    if buf is None or buf.numel() < 128 * cout:
        buf = _STATS[key] = torch.zeros(128 * max(cout, 1024), device=device)
    return buf[:128 * cout]          # (32 slots, 2, cout) fp64 accumulators seen as floats by the allocator (include/dpmn_hip.h)


def _bn_finalize(stats, bn, count):
    Cc = bn.weight.shape[0]
    dev = stats.device
    scale, shift, mean, rstd = torch.empty(4, Cc, device=dev).unbind(0)
    nbt = bn.num_batches_tracked
    check(lib.dpmn_bn_finalize_f32(dptr(stats), dptr(bn.weight), dptr(bn.bias), float(count), float(bn.eps), float(bn.momentum),
                                   dptr(scale), dptr(shift), dptr(mean), dptr(rstd), dptr(bn.running_mean), dptr(bn.running_var),
                                   Cc, nbt.data_ptr() if nbt is not None and nbt.is_cuda else None, 1, stream()))
    return scale, shift, mean, rstd


class Unit:
    """One convolution of the CMM graph with its (optional) BatchNorm."""

    def __init__(self, kind, conv, bn, inputs, pro_act, cin_pad=None):
        self.kind, self.conv, self.bn, self.inputs, self.pro_act, self.cin_pad = kind, conv, bn, inputs, pro_act, cin_pad
        self.out = None

    # geometry of the forward conv on the packed weights
    def _fw(self):
        k = self.kind
        if k == "conv3":
            return dict(k=3, stride=1, pad=1, dil=1)
        if k == "conv4s2d2":
            return dict(k=4, stride=2, pad=3, dil=2)
        if k == "conv4s2":
            return dict(k=4, stride=2, pad=1, dil=1)
        if k == "convT3":
            return dict(k=3, stride=1, pad=1, dil=1)
        raise ValueError(k)

    def forward(self):
        w, b = self.conv.weight, self.conv.bias
        xs = [t.r for t in self.inputs]
        aff = [t.aff for t in self.inputs]
        transposed = self.kind in ("convT3", "convT4s2")
        cout = w.shape[1] if transposed else w.shape[0]
        stats = stats_buffer(cout, w.device) if self.bn is not None else None
        if self.kind == "convT4s2":
            r = ops.convT_s2k4(xs, (packing.tpack_convT_s2k4(w), b), cout, pro_act=self.pro_act, affine=aff, stats=stats)
        else:
            wp = packing.tpack_convT_s1(w) if self.kind == "convT3" else packing.tpack_conv(w, cin_pad=self.cin_pad)
            r = ops.conv2d(xs, wp, b, cout, pro_act=self.pro_act, affine=aff, stats=stats, **self._fw())
        if self.bn is not None:
            count = r.numel() // cout
            self.out = T(r, self.bn, *_bn_finalize(stats, self.bn, count))
        else:
            self.out = T(r)
        return self.out

    def backward(self, gr, side=None):
        """Consumes self.out.G; accumulates parameter grads into gr; pushes gradients to the inputs' G buffers.
        side: a stream for the weight gradient (the caller joins it before anything reads gr), or None = in stream order."""
        dr, cout = self.bwd_output(gr)
        self.bwd_weight(dr, cout, gr, side)
        self.bwd_inputs(dr, cout)

    def bwd_output(self, gr, dr_out=None):
        """dL/d(raw conv output) from self.out.G: BatchNorm backward through the batch statistics (+ the bias gradient of a conv
        without BatchNorm).  dr_out: where to write it (one half of a twin pair's buffer, backward_pair)."""
        o = self.out
        w, b = self.conv.weight, self.conv.bias
        transposed = self.kind in ("convT3", "convT4s2")
        cout = w.shape[1] if transposed else w.shape[0]
        G = o.G
        pixels = G.numel() // cout
        if self.bn is not None:
            dr = torch.empty_like(G) if dr_out is None else dr_out
            ws = torch.empty(2, cout, device=G.device)
            if getattr(o, "gsums_done", False):
                # the consumer that completed G already reduced (sum G, sum G * xhat) per channel: no pass over G and r here
                check(lib.dpmn_bn_bwd_apply_f32(dptr(G), dptr(o.r), dptr(self.bn.weight), dptr(o.mean), dptr(o.rstd), o.gsums.data_ptr(), dptr(ws),
                                                dptr(dr), dptr(gr[self.bn.weight]), dptr(gr[self.bn.bias]), pixels, cout, stream()))
            else:
                check(lib.dpmn_bn_bwd_f32(dptr(G), dptr(o.r), dptr(self.bn.weight), dptr(o.mean), dptr(o.rstd), dptr(ws), dptr(dr),
                                          dptr(gr[self.bn.weight]), dptr(gr[self.bn.bias]), pixels, cout, stream()))
        else:
            dr = G
        cout_real = cout
        if cout % 4 != 0:   # de_1 (Cout = 3): pad the output-gradient channels so NHWC rows stay 16-byte aligned
            cout = (cout + 3) // 4 * 4
            dr = torch.cat([dr, dr.new_zeros(*dr.shape[:3], cout - cout_real)], dim=3)
            tmp = torch.zeros(cout, device=dr.device)
            colsum(dr.reshape(pixels, cout), tmp)
            gr[b] += tmp[:cout_real]
        elif self.bn is None:
            colsum(dr.reshape(pixels, cout), gr[b])
        # else: a conv bias in front of a train-mode BatchNorm has an identically zero gradient (the batch mean removes it;
        # sum over pixels of dr is exactly 0 in exact arithmetic) -- autograd in the reference only accumulates round-off
        # there (<= 3e-4 at these sizes), so the bias gradient stays at its zero fill instead of costing a reduction
        return dr, cout

    def bwd_weight(self, dr, cout, gr, side=None):
        w = self.conv.weight
        xs = [t.r for t in self.inputs]
        aff = [t.aff for t in self.inputs]
        # ---- weight gradient: a leaf of the backward graph (nothing downstream reads it before the optimizer), so it runs on a
        # side stream beside the data-gradient chain that the next unit waits for; backward() joins the streams at its end
        def wgrad():
            if self.kind == "convT4s2":
                for py in range(2):
                    for px in range(2):
                        d = ops.conv_desc(xs, 2, cout=cout, pro_act=self.pro_act, affine=aff, phase=(py, px))
                        conv_wgrad_into(d, dr, gr[w], "convT_s2k4", (py, px))
            else:
                f = self._fw()
                d = ops.conv_desc(xs, f["k"], f["stride"], f["pad"], f["dil"], cout=cout, pro_act=self.pro_act, affine=aff)
                conv_wgrad_into(d, dr, gr[w], "convT_s1" if self.kind == "convT3" else "conv")
        if side is None:
            wgrad()
        else:
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                wgrad()
            dr.record_stream(side)       # dr dies with this call: its memory must not be reused before the side stream read it
            for t in self.inputs:        # (the saved activations die with the graph at the end of the backward, which no longer waits
                t.r.record_stream(side)  #  for the side stream: LAZY_WGRAD_JOIN)
                if t.scale is not None:
                    t.scale.record_stream(side)

    def bwd_inputs(self, dr, cout):
        # ---- data gradients, one launch per input segment, then activation backward through the producer's affine
        w = self.conv.weight
        dev = dr.device
        c0 = 0
        for t in self.inputs:
            cs = t.r.shape[3]
            if t.G is False:       # leaf that needs no gradient
                c0 += cs
                continue
            B, Hi, Wi, _ = t.r.shape
            if self.kind == "conv3":
                wt = packing.tpack_dgrad_conv_s1(w, c0, cs)
                dA = ops.conv2d([dr], wt, None, cs, 3, pad=1)
            elif self.kind == "convT3":
                wt = packing.tpack_dgrad_convT(w, c0, cs, cin_pad=cout)   # (Cin_seg, Cout, 3, 3) seen as Conv2d (out=Cin_seg, in=Cout)
                dA = ops.conv2d([dr], wt, None, cs, 3, pad=1)
            elif self.kind == "convT4s2":
                wt = packing.tpack_dgrad_convT(w, c0, cs)
                dA = ops.conv2d([dr], wt, None, cs, 4, stride=2, pad=1)
            elif self.kind == "conv4s2":
                # Conv2d weight (Cout, Cin, 4, 4) seen as ConvTranspose (in=Cout, out=Cin)
                dA = ops.convT_s2k4([dr], (packing.tpack_dgrad_conv_s2k4(w, c0, cs), None), cs)
            elif self.kind == "conv4s2d2":
                # iy = 2*oy + 2*ky - 3 is always odd: dX[2s+1] = sum_ky dY[s + 2 - ky] W[ky]; even pixels get zero gradient
                wt = packing.tpack_dgrad_generic(w, c0, cs)
                dA = torch.zeros(B, Hi, Wi, cs, device=dev)
                Ho, Wo = dr.shape[1], dr.shape[2]
                geom = dict(stride=1, dil_y=-1, dil_x=-1, pad_y=-2, pad_x=-2, Hp=Hi // 2, Wp=Wi // 2, Hout=Hi, Wout=Wi, ostep=2, ooy=1, oox=1)
                ops.conv2d([dr], wt, None, cs, 4, out=dA, geom=geom)
            else:
                raise ValueError(self.kind)
            self.push_grad(t, dA)
            c0 += cs

    def push_grad(self, t, dA):
        """Activation backward through the producer's affine: t.G (+)= dA * act'(scale * r + shift)."""
        cs = t.r.shape[3]
        sc, sh = (t.scale, t.shift) if t.scale is not None else (None, None)
        first = t.G is None          # first consumer writes, later ones accumulate: no zero fill of the activation
        if first:
            t.G = torch.empty_like(t.r)
        t.n_done = getattr(t, "n_done", 0) + 1
        if (getattr(t, "gsums", None) is not None and t.n_done == t.n_cons and cs % 4 == 0 and 256 % (cs // 4) == 0
                and self.pro_act in ("none", "relu", "leaky02", "leaky001")):
            # last consumer of a BatchNorm output: G is complete after this call -- fold the producer's BatchNorm-backward
            # reduction into it (csrc/conv_bwd.hip k_affine_act_bwd_stats)
            check(lib.dpmn_affine_act_bwd_stats_f32(dptr(dA), dptr(t.r), dptr(sc, True), dptr(sh, True), ACT[self.pro_act], dptr(t.G),
                                                    0 if first else 1, t.r.numel() // cs, cs, dptr(t.mean), dptr(t.rstd), t.gsums.data_ptr(),
                                                    stream()))
            t.gsums_done = True
        else:
            check(lib.dpmn_affine_act_bwd_f32(dptr(dA), dptr(t.r), dptr(sc, True), dptr(sh, True), ACT[self.pro_act], dptr(t.G),
                                              0 if first else 1, t.r.numel() // cs, cs, stream()))


# 1 (default): the data gradients of the twin encoder branches (cmm.py:86-99: same geometry, own weights) as ONE grouped launch per
# level (groups = 2 of dpmn_conv2d_nhwc_f32, the launch the eval forward uses for the twins) instead of two half-batch launches with
# twice the split-K; 0: every unit on its own
GROUP_BWD = os.environ.get("DPMN_CMM_GROUP_BWD", "1") != "0"
GROUP_BWD_MAXPIX = int(os.environ.get("DPMN_CMM_GROUP_BWD_MAXPIX", "12288"))      # pair only levels with at most this many pixels per branch


def _pair_pack(fn, u1, u2, cs):
    """(2, cs, Kp): the two branches' data-gradient packs in ONE buffer (w_group_stride apart), kept on the first branch's conv module
    so that the step's multi-descriptor pack launch (model/packing.py PackCache) refreshes it in place."""
    w1, w2 = u1.conv.weight, u2.conv.weight
    store = u1.conv.__dict__.setdefault("_dpmn_pair_packs", {})
    key = (fn.__name__, cs, w1.data_ptr(), w2.data_ptr())
    buf = store.get(key)
    if buf is None:
        o, _i, kh, kw = w1.shape
        store.clear()       # (a new Trainer moved the parameters: drop the buffers of the old addresses)
        buf = store[key] = torch.empty(2, cs, (kh * kw * o + 31) // 32 * 32, device=w1.device)
    p1, p2 = fn(w1, 0, cs, out=buf[0]), fn(w2, 0, cs, out=buf[1])
    if p1.data_ptr() != buf[0].data_ptr() or p2.data_ptr() != buf[1].data_ptr():
        return torch.stack([p1, p2])      # the pack cache already held separate tensors for these weights
    return buf


def pair_ok(u1, u2):
    """Can the data gradients of these two twin units share a grouped launch?  Same kind and shapes, one input each with >= 32
    channels (the 4-channel leaf of en_1 runs on the Cout <= 4 kernel, which has no grouped form), both inputs want a gradient, the
    BatchNorm backward writes dL/d(raw output) of each branch into its half of one buffer, and the pixels of one half fill whole row
    tiles of the grouped launch."""
    if not GROUP_BWD or u1.kind != u2.kind or u1.kind not in ("conv3", "conv4s2d2") or len(u1.inputs) != 1 or len(u2.inputs) != 1:
        return False
    t1, t2 = u1.inputs[0], u2.inputs[0]
    if t1.G is False or t2.G is False or t1.r.shape != t2.r.shape or t1.r.shape[3] < 32 or u1.out.r.shape != u2.out.r.shape:
        return False
    if u1.out.G is None or u2.out.G is None or u1.conv.weight.shape != u2.conv.weight.shape or (u1.bn is None) != (u2.bn is None):
        return False
    B, Hi, Wi, _ = t1.r.shape
    pix = B * Hi * Wi if u1.kind == "conv3" else B * (Hi // 2) * (Wi // 2)
    if pix % (64 if pix <= 512 else 128) != 0 or pix > GROUP_BWD_MAXPIX:
        return False
    return u1.bn is not None      # (a unit without BatchNorm hands its G on as dr: the halves would have to share a buffer already)


def backward_pair(u1, u2, gr, side=None):
    """Unit.backward of two twin encoder units with the data gradient as one grouped launch."""
    B = u1.out.r.shape[0]
    drp = torch.empty(2 * B, *u1.out.r.shape[1:], device=u1.out.r.device)      # the BatchNorm backward of each branch fills its half
    dr1, cout = u1.bwd_output(gr, drp[:B])
    dr2, _ = u2.bwd_output(gr, drp[B:])
    u1.bwd_weight(dr1, cout, gr, side)
    u2.bwd_weight(dr2, cout, gr, side)
    t1, t2 = u1.inputs[0], u2.inputs[0]
    _, Hi, Wi, cs = t1.r.shape
    if u1.kind == "conv3":
        dAp = ops.conv2d([drp], _pair_pack(packing.tpack_dgrad_conv_s1, u1, u2, cs), None, cs, 3, pad=1, groups=2)
    else:       # conv4s2d2: the odd-pixel scatter of Unit.bwd_inputs
        dAp = torch.zeros(2 * B, Hi, Wi, cs, device=drp.device)
        geom = dict(stride=1, dil_y=-1, dil_x=-1, pad_y=-2, pad_x=-2, Hp=Hi // 2, Wp=Wi // 2, Hout=Hi, Wout=Wi, ostep=2, ooy=1, oox=1)
        ops.conv2d([drp], _pair_pack(packing.tpack_dgrad_generic, u1, u2, cs), None, cs, 4, out=dAp, geom=geom, groups=2)
    u1.push_grad(t1, dAp[:B])
    u2.push_grad(t2, dAp[B:])


WGRAD_STREAM = os.environ.get("DPMN_WGRAD_STREAM", "1") != "0"
LAZY_WGRAD_JOIN = os.environ.get("DPMN_LAZY_WGRAD_JOIN", "1") != "0"      # 0: the backward's stream waits for the weight-gradient stream at its end
FUSE_BN_REDUCE = os.environ.get("DPMN_BN_BWD_FUSED", "1") != "0"      # 0: dpmn_bn_bwd_f32 with its own reduction pass
_SIDE = {}


def wgrad_stream(dev):
    """The side stream of the CMM weight gradients (one per device), or None: switched off, or the step is being captured into
    a hipGraph (graphed_train_step keeps the single-stream order)."""
    if not WGRAD_STREAM or (torch.cuda.is_current_stream_capturing() and os.environ.get("DPMN_GRAPH_MULTISTREAM", "1") == "0"):
        return None
    from .. import _streams
    return _streams.pool(dev)["wgrad"]


def build(m, x1, x2):
    """Forward in training mode; returns (out NCHW, graph) where graph lists the units in execution order."""
    units = []
    # a train-mode forward moves the BatchNorm running statistics through raw pointers (no torch _version bump), so the
    # eval-mode pack of model/cmm.py (folded BatchNorm) is stale from here on
    m._pack = None

    def run(kind, conv, bn, inputs, act, cin_pad=None):
        u = Unit(kind, conv, bn, inputs, act, cin_pad)
        units.append(u)
        return u.forward()

    enc, leaves, enc_units = [], [], []
    for br, x in (("1", x1), ("2", x2)):
        n_before = len(units)
        leaf = T(ops.nchw_to_nhwc(x.contiguous().float(), 4))
        leaves.append(leaf)
        o = [run("conv3", getattr(m, "en_1_" + br), None, [leaf], "none", cin_pad=4)]
        for lvl in (2, 3, 4, 5):
            seq = getattr(m, "en_%d_%s" % (lvl, br)).encode
            t = run("conv4s2d2", seq[1], seq[2], [o[-1]], "leaky02")
            o.append(run("conv3", seq[4], seq[5], [t], "leaky02"))
        o.append(run("conv4s2", getattr(m, "en_6_" + br)[1], None, [o[-1]], "leaky02"))
        enc.append(o)
        enc_units.append(units[n_before:])
    a, b = enc
    bott = torch.cat([a[5].r, b[5].r], dim=3)
    gated = T(ops.se_gate(bott, m.fc_1.weight, m.fc_1.bias, m.fc_2.weight, m.fc_2.bias))
    d = run("convT4s2", m.de_6[1], m.de_6[2], [gated], "relu")
    for lvl, skip in ((5, 4), (4, 3), (3, 2), (2, 1)):
        seq = getattr(m, "de_%d" % lvl).decode
        t = run("convT3", seq[1], seq[2], [d, a[skip], b[skip]], "relu")
        d = run("convT4s2", seq[4], seq[5], [t], "relu")
    last = Unit("convT3", m.de_1[1], None, [d, a[0], b[0]], "relu")
    units.append(last)
    wp, bp = packing.tpack_convT_s1(m.de_1[1].weight), m.de_1[1].bias
    out = ops.conv2d([d.r, a[0].r, b[0].r], wp, bp, m.c_img, 3, pad=1, pro_act="relu", affine=[d.aff, a[0].aff, b[0].aff], out_nchw=True)
    last.out = T(None)
    return out, dict(units=units, bott=bott, gated=gated, enc=enc, leaves=leaves, last=last, twins=list(zip(*enc_units)))


def backward(m, graph, dout, need_dx=(True, True)):
    gr, direct = grad_targets(m)
    units, last = graph["units"], graph["last"]
    a, b = graph["enc"]
    for leaf, need in zip(graph["leaves"], need_dx):
        if not need:
            leaf.G = False
    # de_1: output was stored NCHW with 3 channels -> NHWC for the backward kernels
    B, Cc, H, W = dout.shape
    last.out.G = dout.permute(0, 2, 3, 1).contiguous()     # layout plumbing of a (B,3,32,128) tensor
    last.out.r = last.out.G
    # consumers per tensor + one zero-filled fp64 slab for the (sum G, sum G * xhat) pairs that the last consumer of every
    # BatchNorm output accumulates (Unit.backward)
    if FUSE_BN_REDUCE:
        for u in units:
            for t in u.inputs:
                t.n_cons = getattr(t, "n_cons", 0) + 1
        bn_outs = [u.out for u in units if u.bn is not None and u.out.mean is not None]
        slab = torch.zeros(sum(2 * t.r.shape[3] for t in bn_outs), dtype=torch.float64, device=dout.device)
        off = 0
        for t in bn_outs:
            n = 2 * t.r.shape[3]
            t.gsums, off = slab[off:off + n], off + n
    partner, second = {}, set()
    if GROUP_BWD:
        for u1, u2 in graph.get("twins", ()):
            partner[id(u2)], partner[id(u1)] = u1, u2
            second.add(id(u2))
    side = wgrad_stream(dout.device)
    # gradient exchange in segments (train/optim.py SegmentedBucket; only when collectives run): how many units of each segment are
    # still to come -- the segment is reported the moment its last unit (and, for the segment that holds fc_1 / fc_2, the channel
    # gate) has issued its gradients, with an event that covers this stream AND the weight-gradient stream
    seg = getattr(m, "_dpmn_bucket", None) if direct else None
    seg = seg if seg is not None and hasattr(seg, "segment_ready") else None
    seg_left = {}
    if seg is not None:
        for u in units:
            k_ = seg.seg_of[id(u.conv.weight)]
            seg_left[k_] = seg_left.get(k_, 0) + 1
        k_gate = seg.seg_of[id(m.fc_1.weight)]
        seg_left[k_gate] = seg_left.get(k_gate, 0) + 1

    def seg_done(key_param):
        if seg is None:
            return
        k_ = seg.seg_of[id(key_param)]
        seg_left[k_] -= 1
        if seg_left[k_] > 0:
            return
        cur_ = torch.cuda.current_stream(dout.device)
        if side is not None and not torch.cuda.is_current_stream_capturing():
            if uq_holder[0] is not None:
                with torch.cuda.stream(side):
                    uq_holder[0].flush()        # this segment's conv weight gradients into the parameter layout
            here = torch.cuda.Event()
            here.record(cur_)
            side.wait_event(here)
            ev = torch.cuda.Event()
            ev.record(side)
            seg.segment_ready(k_, ev)
        else:
            if uq_holder[0] is not None:
                uq_holder[0].flush()
            seg.segment_ready(k_)
    uq_holder = [None]
    # every conv's weight-gradient unpack in ONE launch at the end (train/pgrm_train.py UnpackQueue; per module: the descriptor
    # table holds this module's gradient sinks)
    from . import pgrm_train as _pt
    uq = None
    if _pt.UNPACK_MULTI and direct and not torch.cuda.is_current_stream_capturing():
        uq = getattr(m, "_unpack_queue", None)
        if uq is None:
            uq = m._unpack_queue = _pt.UnpackQueue()
        _pt.UNPACK_QUEUE = uq
    uq_holder[0] = uq
    try:
      # reverse execution order; the units of the second encoder branch wait for their twins (branch 1 comes later in this order): a
      # twin pair runs when its FIRST-branch unit is reached, after both units' consumers (decoder, next encoder level) are done
      waiting = set()
      for u in reversed(units):
          tw = partner.get(id(u))
          if tw is not None and id(u) in second:
              waiting.add(id(u))      # branch-2 unit: deferred until its branch-1 twin comes up
              continue
          if tw is not None and id(tw) in waiting:
              if tw.out.G is not None and u.out.G is not None and pair_ok(u, tw):
                  backward_pair(u, tw, gr, side)
              else:
                  if tw.out.G is not None:
                      tw.backward(gr, side)
                  if u.out.G is not None:
                      u.backward(gr, side)
              seg_done(tw.conv.weight)
              seg_done(u.conv.weight)
              continue
          if u is last or u.out.G is not None:
              u.backward(gr, side)
          seg_done(u.conv.weight)
          if u.inputs and u.inputs[0] is graph["gated"]:
              # channel gate backward, then split the bottleneck gradient into the two en_6 outputs
              g = graph["gated"]
              dbott = torch.empty_like(graph["bott"])
              _c, _cm = dbott.shape[3], m.fc_1.weight.shape[0]
              se_ws = torch.empty(dbott.shape[0] * (5 * _c + 2 * _cm) + 2 * _c * _cm, device=dbott.device)
              check(lib.dpmn_se_gate_bwd_f32(dptr(graph["bott"]), dptr(g.G), dptr(m.fc_1.weight), dptr(m.fc_1.bias), dptr(m.fc_2.weight),
                                             dptr(m.fc_2.bias), dptr(dbott), dptr(gr[m.fc_1.weight]), dptr(gr[m.fc_1.bias]),
                                             dptr(gr[m.fc_2.weight]), dptr(gr[m.fc_2.bias]), dptr(se_ws), dbott.shape[0], dbott.shape[1] * dbott.shape[2],
                                             dbott.shape[3], m.fc_1.weight.shape[0], stream()))
              half = dbott.shape[3] // 2
              a[5].G = dbott[..., :half].contiguous()
              b[5].G = dbott[..., half:].contiguous()
              seg_done(m.fc_1.weight)
    finally:
        _pt.UNPACK_QUEUE = None
    if uq is not None:
        if side is not None:
            with torch.cuda.stream(side):
                uq.flush()
        else:
            uq.flush()
    dxs = []
    for leaf, need in zip(graph["leaves"], need_dx):
        dxs.append(ops.nhwc_to_nchw(leaf.G)[:, :3].contiguous() if need else None)
    if side is not None:
        cur = torch.cuda.current_stream(dout.device)
        if direct and LAZY_WGRAD_JOIN and getattr(m._dpmn_bucket, "lazy_join", False) and not torch.cuda.is_current_stream_capturing():
            # the data gradients go on to the PGRM backwards on THIS stream; the weight gradients (and their 0.4 ms unpack launch at the
            # end of the side stream) are only needed by the optimizer / the gradient exchange: the bucket gets an event that covers both
            # streams (FlatBucket._mark_ready) instead of this stream waiting here
            here = torch.cuda.Event()
            here.record(cur)
            side.wait_event(here)
            done = torch.cuda.Event()
            done.record(side)
            m._dpmn_grads_done = done
        else:
            cur.wait_stream(side)      # every weight gradient is in place before the gradients are handed to autograd / the bucket
    return dxs, gr, direct


class CMMFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, x1, x2, *params):
        out, graph = build(m, x1, x2)
        ctx.m, ctx.graph = m, graph
        ctx.need = (x1.requires_grad, x2.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        dxs, gr, direct = backward(ctx.m, ctx.graph, dout.contiguous(), ctx.need)
        ctx.graph = None
        return (None, dxs[0], dxs[1]) + finish_grads(ctx.m, gr, direct)


def apply(m, x1, x2):
    if getattr(m, "_dpmn_bucket", None) is not None:
        m._dpmn_bucket.note_use()
    return CMMFunction.apply(m, x1, x2, *fn_inputs(m))

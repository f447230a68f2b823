"""GPU parity: BiGRU / TPInterpreter kernels and the TSRN / TATT PSN modules vs oracle and golden."""
import pytest
import torch
import torch.nn.functional as F

from dpmn_amd.utils import synth
from helpers import load_golden, t, assert_close

pytestmark = pytest.mark.gpu
ATOL, RTOL = 1e-4, 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def u(name, shape, lo=-1.0, hi=1.0, seed=60):
    return synth.uniform(name, shape, lo, hi, seed)


@pytest.mark.parametrize("axis", ["w", "h"])
def test_bigru_matches_oracle(dev, axis):
    from dpmn_amd import ops
    from oracle import tsrn as ot
    B, H, W, C = 3, 16, 64, 64
    sd = {}
    for sfx in ("", "_reverse"):
        sd["weight_ih_l0" + sfx] = u("wih" + sfx, (96, C), -0.3, 0.3)
        sd["weight_hh_l0" + sfx] = u("whh" + sfx, (96, 32), -0.4, 0.4)
        sd["bias_ih_l0" + sfx] = u("bih" + sfx, (96,), -0.2, 0.2)
        sd["bias_hh_l0" + sfx] = u("bhh" + sfx, (96,), -0.2, 0.2)
    x = u("x", (B, H, W, C))
    res = u("res", (B, H, W, C))
    seqs = x.reshape(B * H, W, C) if axis == "w" else x.permute(0, 2, 1, 3).reshape(B * W, H, C)
    ref = ot.bigru(seqs, sd, "")
    ref = ref.reshape(B, H, W, C) if axis == "w" else ref.reshape(B, W, H, C).permute(0, 2, 1, 3)
    wi = torch.cat([sd["weight_ih_l0"], sd["weight_ih_l0_reverse"]], 0)
    bi = torch.cat([sd["bias_ih_l0"], sd["bias_ih_l0_reverse"]], 0)
    gi = F.linear(x, wi, bi)
    whh = torch.stack([sd["weight_hh_l0"], sd["weight_hh_l0_reverse"]], 0)
    bhh = torch.stack([sd["bias_hh_l0"], sd["bias_hh_l0_reverse"]], 0)
    got = ops.bigru(gi.to(dev).contiguous(), whh.to(dev), bhh.to(dev), B, H, W, axis, res=res.to(dev))
    assert_close(got, ref + res, ATOL, RTOL, "bigru axis " + axis)


def test_small_linear_layernorm_cross_attn(dev):
    from dpmn_amd import ops
    from oracle import tsrn as ot
    x, add = u("x", (52, 37)), u("add", (26, 37))
    w, b = u("w", (64, 37), -0.3, 0.3), u("b", (64,))
    ref = F.prelu(F.linear(x + add.repeat(2, 1), w, b), torch.tensor([0.2]))
    assert_close(ops.small_linear(x.to(dev), w.to(dev), b.to(dev), add=add.to(dev), act="prelu", slope=0.2), ref, ATOL, RTOL, "small_linear")
    # ragged tiles of the LDS-staged kernel (32-row x 64-column tiles, 64-deep K chunks): the interpreter's 1248 x 64 x 64 shape, a row
    # / column / K remainder each, K above one chunk, no bias, no addend
    for (M, N, K, rows, bias) in ((1248, 64, 64, 26, True), (1251, 70, 37, 0, True), (33, 130, 150, 11, False), (5, 3, 64, 0, True)):
        x2, w2 = u("x2%d" % M, (M, K)), u("w2%d" % M, (N, K), -0.3, 0.3)
        b2 = u("b2%d" % M, (N,)) if bias else None
        add2 = u("add2%d" % M, (rows, K)) if rows else None
        xin = x2 if add2 is None else x2 + add2.repeat((M + rows - 1) // rows, 1)[:M]
        ref2 = F.linear(xin, w2, b2)
        got2 = ops.small_linear(x2.to(dev), w2.to(dev), None if b2 is None else b2.to(dev), add=None if add2 is None else add2.to(dev))
        assert_close(got2, ref2, ATOL, RTOL, "small_linear %dx%dx%d" % (M, N, K))
    a, r = u("a", (300, 64)), u("r", (300, 64))
    g1, b1, g2, b2 = u("g1", (64,), 0.5, 1.5), u("b1", (64,)), u("g2", (64,), 0.5, 1.5), u("b2", (64,))
    y = F.layer_norm(a + r, (64,), g1, b1)
    acc = torch.full((300, 64), 0.25)
    accd = acc.to(dev)
    got = ops.add_layernorm64(a.to(dev), r.to(dev), g1.to(dev), b1.to(dev), g2.to(dev), b2.to(dev), accd, alpha=0.5, accumulate=True)
    assert_close(got, y, ATOL, RTOL, "add_layernorm")
    assert_close(accd, acc + 0.5 * F.layer_norm(y, (64,), g2, b2), ATOL, RTOL, "add_layernorm accumulate")
    N, L, S, E = 3, 1024, 26, 64
    q, k, v = u("q", (N, L, E), -2, 2), u("k", (N, S, E), -2, 2), u("v", (N, S, E))
    qh = q.reshape(N, L, 4, 16).permute(0, 2, 1, 3)
    kh = k.reshape(N, S, 4, 16).permute(0, 2, 1, 3)
    vh = v.reshape(N, S, 4, 16).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / 4.0, -1)
    ref_o = (p @ vh).permute(0, 2, 1, 3).reshape(N, L, E)
    o, pw = ops.cross_attn(q.to(dev), k.to(dev), v.to(dev), need_weights=True)
    assert_close(o, ref_o, ATOL, RTOL, "cross_attn out")
    assert_close(pw, p.mean(1), 1e-5, 1e-4, "cross_attn weights")


def _load(cls, name, seed, dev, **kw):
    g = load_golden(name)
    m = cls(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32, **kw).eval()
    sd = m.state_dict()
    assert list(sd.keys()) == [str(r).split("|")[0] for r in g["manifest"]], "state_dict key layout differs from the reference"
    synth.synth_fill_(sd, seed)
    m.load_state_dict(sd)
    return g, m.to(dev)


def test_tsrn_module_vs_reference_golden(dev):
    from dpmn_amd.model.tsrn import TSRN
    g, m = _load(TSRN, "tsrn", 41, dev)
    x = synth.synth_batch(2, seed=2)["images_lr"].to(dev)
    with torch.no_grad():
        out = m(x)
    assert_close(out, t(g["out"]), 2e-4, 2e-4, "TSRN vs reference golden")


@pytest.mark.parametrize("cache", [True, False])
def test_tatt_module_vs_reference_golden(dev, cache):
    from dpmn_amd.model.tatt import TSRN_TL_TRANS
    g, m = _load(TSRN_TL_TRANS, "tatt", 42, dev, cache_query_embed=cache)
    b = synth.synth_batch(2, seed=2)
    with torch.no_grad():
        for _ in range(2):   # second call exercises the cached query embedding
            out, prw = m(b["images_lr"].to(dev), b["label_vecs"].to(dev))
    assert_close(out, t(g["out"]), 2e-4, 2e-4, "TATT vs reference golden")
    assert_close(prw[:, ::16], t(g["pr_weights"]), 1e-5, 1e-4, "TATT pr_weights vs reference golden")


def test_tatt_batch48_vs_oracle(dev):
    from dpmn_amd.model.tatt import TSRN_TL_TRANS
    from oracle import tsrn as ot
    _, m = _load(TSRN_TL_TRANS, "tatt", 43, dev)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = synth.synth_batch(48, seed=3)
    ref, _ = ot.tatt_forward(sd, b["images_lr"], b["label_vecs"])
    with torch.no_grad():
        out, _ = m(b["images_lr"].to(dev), b["label_vecs"].to(dev))
    assert_close(out, ref, 2e-4, 2e-4, "TATT B=48 vs oracle")


def test_tpgsr_module_vs_reference_golden(dev):
    """--arch tpgsr: TSRN_TL (tsrn.py:153-247) eval forward vs the imported reference class (tests/golden/tpgsr.npz)."""
    from dpmn_amd.model.tsrn import TSRN_TL
    g, m = _load(TSRN_TL, "tpgsr", 44, dev)
    b = synth.synth_batch(2, seed=2)
    with torch.no_grad():
        out = m(b["images_lr"].to(dev), b["label_vecs"].to(dev))
        P = m._trunk_pack()
        info = m._info_gen(b["label_vecs"].to(dev), 1, 203, P)      # a 203-column target = the InfoGen map itself
    assert_close(out, t(g["out"]), 2e-4, 2e-4, "TSRN_TL vs reference golden")
    assert_close(info[:, 0, ::7, :].permute(0, 2, 1), t(g["info"]), 2e-5, 2e-5, "InfoGen map vs reference golden")


def test_tpgsr_batch48_vs_oracle_and_native_trunk(dev):
    from dpmn_amd.model import tsrn as tsrn_mod
    from oracle import tsrn as ot
    _, m = _load(tsrn_mod.TSRN_TL, "tpgsr", 45, dev)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = synth.synth_batch(48, seed=3)
    ref = ot.tsrn_tl_forward(sd, b["images_lr"], b["label_vecs"])
    x, lv = b["images_lr"].to(dev), b["label_vecs"].to(dev)
    with torch.no_grad():
        out = m(x, lv)
        tsrn_mod.NATIVE_TRUNK = False
        try:
            composed = m(x, lv)
        finally:
            tsrn_mod.NATIVE_TRUNK = True
    assert_close(out, ref, 2e-4, 2e-4, "TSRN_TL B=48 vs oracle")
    assert torch.equal(out, composed), "native trunk differs from the composed ops"


def test_tpgsr_stack_refine_vs_oracle(dev):
    """`--arch tpgsr` through TextSR.refine (TSRN_TL PSN + 1+1 PGRM + CMM, B = 4) vs the oracle stack."""
    from dpmn_amd import workload
    from dpmn_amd.interfaces.super_resolution import TextSR
    from oracle import dpmn as odpmn
    B = 4
    sr = TextSR(workload.make_config(B), workload.make_args("tpgsr", 1, 1, B))
    models, psn = sr.build_models()
    for i, m in enumerate([psn] + models):
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=140 + i)
        m.load_state_dict(sd)
        m.eval()
    b = synth.synth_batch(B, seed=9)
    pri = [torch.floor(synth.uniform("tpg_prior", (B, 2, 32, 128), 0.0, 256.0, 9))]
    out = sr.refine(models, psn, b["images_lr"].to(dev), b["label_vecs"].to(dev), text_priors=[p.to(dev) for p in pri])
    sds, sd_psn = workload.state_dicts_cpu(models, psn)
    ref = odpmn.refine(sd_psn, sds[:-1], sds[-1], "tpgsr", 1, 1, b["images_lr"], b["label_vecs"], pri, 0.5)
    assert_close(out, ref, 5e-4, 5e-4, "tpgsr stack")


@pytest.mark.parametrize("arch,B", [("tsrn", 2), ("tsrn", 48), ("tatt", 3), ("tatt", 48)])
def test_psn_native_trunk_equals_composed_ops(dev, arch, B):
    """dpmn_psn_trunk_f32 (SRBs + tail from one native call, csrc/psn_forward.hip) issues the launches of the per-op host path
    with the same arguments: bitwise-equal outputs, also on the second call (cached weights struct / workspace)."""
    from dpmn_amd.model import tsrn as tsrn_mod
    from dpmn_amd.model.tatt import TSRN_TL_TRANS
    _, m = _load(tsrn_mod.TSRN if arch == "tsrn" else TSRN_TL_TRANS, arch, 44, dev)
    b = synth.synth_batch(B, seed=4)
    args = (b["images_lr"].to(dev),) + ((b["label_vecs"].to(dev),) if arch == "tatt" else ())
    first = lambda o: o[0] if isinstance(o, tuple) else o
    assert tsrn_mod.NATIVE_TRUNK
    with torch.no_grad():
        nat, nat2 = first(m(*args)), first(m(*args))
        tsrn_mod.NATIVE_TRUNK = False
        try:
            ref = first(m(*args))
        finally:
            tsrn_mod.NATIVE_TRUNK = True
    assert torch.equal(nat, ref) and torch.equal(nat2, ref)


def test_mha32_and_layernorm_std_match_oracle(dev):
    """TBSRN FeatureEnhancer kernels (tbsrn.py:110-150, 23-36) vs plain softmax attention / std LayerNorm."""
    import math
    from dpmn_amd import ops
    B, L, Hh = 3, 1024, 4
    qkv = u("mha_qkv", (B * L, 3 * Hh * 32), -2.0, 2.0)
    q, k, v = (qkv[:, i * 128:(i + 1) * 128].reshape(B, L, Hh, 32).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(32.0), -1) @ v).transpose(1, 2).reshape(B * L, 128)
    assert_close(ops.mha32(qkv.to(dev), B, L, Hh, 1.0 / math.sqrt(32.0)), ref, ATOL, RTOL, "mha32")
    x, a2, b2 = u("lns_x", (777, 128), -3, 3), u("lns_a", (128,), 0.5, 1.5), u("lns_b", (128,))
    ref = a2 * (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + 1e-6) + b2
    assert_close(ops.layernorm_std(x.to(dev), a2.to(dev), b2.to(dev), 1e-6), ref, ATOL, RTOL, "layernorm_std")


def test_tbsrn_module_vs_reference_golden_and_oracle(dev):
    """a14: TBSRN mirror (eval) vs the reference's own output (B=2) and vs the oracle at B=5."""
    from dpmn_amd.model.tbsrn import TBSRN
    from oracle import tsrn as ot
    g, m = _load(TBSRN, "tbsrn", 43, dev)
    x = synth.synth_batch(2, seed=2)["images_lr"].to(dev)
    with torch.no_grad():
        out = m(x)
    assert_close(out, t(g["out"]), 2e-4, 2e-4, "TBSRN vs reference golden")
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    xb = synth.synth_batch(5, seed=4)["images_lr"]
    with torch.no_grad():
        out = m(xb.to(dev))
    assert_close(out, ot.tbsrn_forward(sd, xb), 2e-4, 2e-4, "TBSRN B=5 vs oracle")


def test_psn_with_stn_entries_runs_like_without(dev):
    """--STN models (the README's configuration) carry the tps.* / stn_head.* entries but never execute them in eval:
    same output as the STN=False model on the same trunk weights."""
    from dpmn_amd.model.tatt import TSRN_TL_TRANS
    kw = dict(scale_factor=2, width=128, height=32, mask=True, srb_nums=5, hidden_units=32)
    a = TSRN_TL_TRANS(STN=True, **kw).eval()
    b = TSRN_TL_TRANS(STN=False, **kw).eval()
    sd = a.state_dict()
    synth.synth_fill_(sd, 45)
    a.load_state_dict(sd)
    b.load_state_dict({k: v for k, v in sd.items() if not k.startswith(("tps.", "stn_head."))})
    bt = synth.synth_batch(3, seed=6)
    with torch.no_grad():
        oa, _ = a.to(dev)(bt["images_lr"].to(dev), bt["label_vecs"].to(dev))
        ob, _ = b.to(dev)(bt["images_lr"].to(dev), bt["label_vecs"].to(dev))
    assert torch.equal(oa, ob)

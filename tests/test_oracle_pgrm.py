"""Pin the PGRM oracle against golden vectors produced by the imported reference
(tools/gen_golden.py).  CPU only."""
import pytest
import torch

from dpmn_amd.utils import synth
from oracle import pgrm as opgrm
from helpers import load_golden, sd_from_manifest, checksum, t, assert_close

TOL = 2e-5  # fp32 CPU vs fp32 CPU, different op order


@pytest.mark.parametrize("tag,it,mode", [("mode0_iter0", 0, False), ("mode1_iter2", 2, True)])
def test_pgrm_forward_matches_reference(tag, it, mode):
    g = load_golden("pgrm_" + tag)
    B, _, _, wseed, iseed = [int(v) for v in g["meta"]]
    sd = sd_from_manifest(g["manifest"], wseed)
    # attn_mask buffers are float in the reference manifest; they are derived, not weights
    assert abs(checksum(sd) - float(g["checksum"])) < 1e-6 * max(1.0, abs(float(g["checksum"])))
    if mode:
        x_q = (synth.uniform("x_q", (B, 1, 32, 128), 0, 1, iseed) > 0.5).float().repeat(1, 3, 1, 1)
    else:
        x_q = torch.floor(synth.uniform("x_q", (B, 2, 32, 128), 0, 256, iseed))
    x_kv = synth.uniform("x_kv", (B, 3, 32, 128), 0, 1, iseed)
    res = [synth.uniform("res%d" % i, (B, 3, 32, 128), 0, 1, iseed) for i in range(it)]
    out = opgrm.pgrm_forward(sd, x_q, x_kv, res)
    assert_close(out, t(g["out"]), TOL, 1e-5, "pgrm " + tag)


@pytest.mark.parametrize("tag,shifts", [("shift0", [0, 0, 0]), ("shifted", [1, 2, 4])])
def test_window_attention_matches_reference(tag, shifts):
    import torch.nn.functional as F
    g = load_golden("wattn_" + tag)
    sd = sd_from_manifest(g["manifest"], 21)
    B, H, W, C = 1, 16, 64, 96
    xq = synth.uniform("wa_xq", (B, H, W, C), -1, 1, 6).reshape(B, H * W, C)
    xkv = synth.uniform("wa_xkv", (B, H, W, C), -1, 1, 6).reshape(B, H * W, C)
    q = F.linear(xq, sd["q.weight"], sd["q.bias"])
    kv = F.linear(xkv, sd["kv.weight"], sd["kv.bias"])
    cat = opgrm.window_attention_core(q, kv[..., :C], kv[..., C:], sd, "", H, W, [2, 4, 8], shifts, 2)
    assert_close(cat, t(g["cat"]), TOL, 1e-5, "cat " + tag)
    out = opgrm.sk_fuse(cat, sd, "sknet.", 3)
    assert_close(out, t(g["out"]), TOL, 1e-5, "sk " + tag)


def test_mlp_matches_reference():
    g = load_golden("mlp")
    sd = sd_from_manifest(g["manifest"], 22)
    x = synth.uniform("mlp_x", (2, 1024, 96), -1, 1, 6)
    assert_close(opgrm.mlp(x, sd, ""), t(g["out"]), TOL, 1e-5, "mlp")


def test_basiclayer_stress_matches_reference():
    g = load_golden("basiclayer_stress")
    sd = sd_from_manifest(g["manifest"], 23)
    xq = synth.uniform("bl_xq", (1, 4096, 192), -1, 1, 6)
    xkv = synth.uniform("bl_xkv", (1, 4096, 192), -1, 1, 6)
    o = opgrm.basic_layer(xq, xkv, sd, "", 32, 128, [4, 8, 16], 6)
    assert_close(o[:, ::7], t(g["out"]), 5e-5, 1e-5, "basiclayer stress")

"""Pin the TSRN / TATT (PSN backbone) oracle against reference-generated golden vectors (CPU)."""
from dpmn_amd.utils import synth
from oracle import tsrn as ot
from helpers import load_golden, sd_from_manifest, checksum, t, assert_close


def _sd(name, seed):
    g = load_golden(name)
    sd = sd_from_manifest(g["manifest"], seed)
    assert abs(checksum(sd) - float(g["checksum"])) < 1e-6 * max(1.0, abs(float(g["checksum"])))
    return g, sd


def test_tsrn_matches_reference():
    g, sd = _sd("tsrn", 41)
    x = synth.synth_batch(2, seed=2)["images_lr"]
    assert_close(ot.tsrn_forward(sd, x), t(g["out"]), 2e-5, 1e-5, "tsrn")


def test_tatt_matches_reference():
    g, sd = _sd("tatt", 42)
    b = synth.synth_batch(2, seed=2)
    out, prw = ot.tatt_forward(sd, b["images_lr"], b["label_vecs"])
    assert_close(out, t(g["out"]), 2e-5, 1e-5, "tatt out")
    assert_close(prw[:, ::16], t(g["pr_weights"]), 1e-6, 1e-5, "tatt pr_weights")


def test_tbsrn_matches_reference():
    """a14: TBSRN eval forward + the first block's FeatureEnhancer output (every 8th position)."""
    g, sd = _sd("tbsrn", 43)
    x = synth.synth_batch(2, seed=2)["images_lr"]
    assert_close(ot.tbsrn_forward(sd, x), t(g["out"]), 2e-5, 1e-5, "tbsrn")
    import torch.nn.functional as F
    b1 = F.prelu(F.conv2d(x, sd["block1.0.weight"], sd["block1.0.bias"], padding=4), sd["block1.1.weight"])
    pre = "block2."
    r = ot.mish(ot.bn_eval(F.conv2d(b1, sd[pre + "conv1.weight"], sd[pre + "conv1.bias"], padding=1), sd, pre + "bn1."))
    r = ot.bn_eval(F.conv2d(r, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"], padding=1), sd, pre + "bn2.")
    fe = ot.feature_enhancer(r.reshape(2, 64, 1024), sd, pre + "feature_enhancer.")
    assert_close(fe[:, ::8], t(g["fe_block2"]), 2e-5, 1e-5, "tbsrn feature enhancer")

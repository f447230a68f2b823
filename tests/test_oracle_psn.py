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

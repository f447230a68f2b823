"""Pin the a15 oracle (oracle/stn.py) against golden vectors produced by the imported reference STNHead (train and eval
mode, incl. the BatchNorm running-stat updates) and TPSSpatialTransformer (tools/gen_golden.py gen_stn_fwd).  CPU only."""
import torch

from dpmn_amd.utils import synth
from oracle import stn as ostn
from helpers import load_golden, sd_from_manifest, checksum, t, assert_close

B = 6


def _setup():
    g = load_golden("stn_fwd")
    sd = sd_from_manifest(g["manifest"], 52)
    assert abs(checksum(sd) - float(g["checksum"])) < 1e-6 * max(1.0, abs(float(g["checksum"])))
    return g, sd, synth.uniform("stn_x", (B, 4, 16, 64), 0, 1, 51)


def test_stn_head_train_and_eval_match_reference():
    g, sd, x = _setup()
    for mode in ("train", "eval"):
        feat, ctrl, new = ostn.stn_head_forward(sd, x, mode == "train")
        assert_close(feat, t(g[mode + "_feat"]), 2e-5, 1e-5, "img_feat " + mode)
        assert_close(ctrl, t(g[mode + "_ctrl"]), 2e-5, 1e-5, "ctrl " + mode)
        if mode == "train":
            assert len(new) == 14                     # 7 BatchNorms x (running_mean, running_var)
            for k, v in new.items():
                assert_close(v, t(g["after." + k]), 2e-5, 1e-6, k)


def test_tps_matches_reference():
    from dpmn_amd.model.stn import TPSSpatialTransformer
    g, _, x = _setup()
    tp = TPSSpatialTransformer(output_image_size=(16, 64), num_control_points=20, margins=(0.05, 0.05))
    ctrl = tp.target_control_points[None] + synth.uniform("stn_ctrl", (B, 20, 2), -0.12, 0.12, 53)
    out, src = ostn.tps_forward(tp.inverse_kernel, tp.target_coordinate_repr, (16, 64), x, ctrl)
    lo, hi = [float(v) for v in g["tps_src_minmax"]]
    assert lo < 0.0 and hi > 1.0, "fixture must exercise the clamp"
    assert_close(src[:, ::7], t(g["tps_src"]), 2e-5, 2e-5, "source coordinates")
    assert_close(out, t(g["tps_out"]), 5e-4, 5e-4, "warped image")

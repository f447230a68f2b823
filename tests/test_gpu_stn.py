"""GPU parity of row a15 (STNHead / TPSSpatialTransformer forward through libdpmn_hip.so) against the golden vectors of the
imported reference and against the CPU oracle at other batch sizes.  fp32; tolerances per assert."""
import pytest
import torch

from dpmn_amd.utils import synth
from helpers import load_golden, sd_from_manifest, t, assert_close

pytestmark = pytest.mark.gpu
B = 6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _head(dev, sd):
    from dpmn_amd.model.stn import STNHead
    m = STNHead(in_planes=4, num_ctrlpoints=20, activation='none')
    m.load_state_dict(sd)
    return m.to(dev)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_stn_head_matches_reference_golden(dev, mode):
    g = load_golden("stn_fwd")
    sd = sd_from_manifest(g["manifest"], 52)
    x = synth.uniform("stn_x", (B, 4, 16, 64), 0, 1, 51)
    m = _head(dev, sd).train(mode == "train")
    feat, ctrl = m(x.to(dev))
    assert ctrl.shape == (B, 20, 2)
    assert_close(feat, t(g[mode + "_feat"]), 1e-4, 1e-4, "img_feat " + mode)
    assert_close(ctrl, t(g[mode + "_ctrl"]), 1e-4, 1e-4, "ctrl " + mode)
    after = m.state_dict()
    for k in after:
        if "running" in k:
            want = t(g["after." + k]) if mode == "train" else sd[k]
            assert_close(after[k], want, 1e-4, 1e-5, k)
        if "num_batches" in k:
            assert int(after[k]) == int(sd[k]) + (1 if mode == "train" else 0), k


def test_stn_head_other_batches_vs_oracle(dev):
    from oracle import stn as ostn
    g = load_golden("stn_fwd")
    sd = sd_from_manifest(g["manifest"], 52)
    # BatchNorm over a handful of samples divides by a tiny variance wherever two pre-activations nearly coincide, which
    # amplifies fp32 summation-order differences: the small batch gets a looser bound
    for b in (4, 17, 48):
        x = synth.uniform("stn_xb", (b, 4, 16, 64), 0, 1, 60 + b)
        for training in (True, False):
            m = _head(dev, sd).train(training)
            feat, ctrl = m(x.to(dev))
            rf, rc, _ = ostn.stn_head_forward(sd, x, training)
            tol = 5e-3 if (b < 8 and training) else 2e-4
            assert_close(feat, rf, tol, tol, "img_feat B=%d train=%s" % (b, training))
            assert_close(ctrl, rc, tol, tol, "ctrl B=%d train=%s" % (b, training))


def test_tps_matches_reference_golden_and_oracle(dev):
    from dpmn_amd.model.stn import TPSSpatialTransformer
    from oracle import stn as ostn
    g = load_golden("stn_fwd")
    x = synth.uniform("stn_x", (B, 4, 16, 64), 0, 1, 51)
    tp = TPSSpatialTransformer(output_image_size=(16, 64), num_control_points=20, margins=(0.05, 0.05))
    ctrl = tp.target_control_points[None] + synth.uniform("stn_ctrl", (B, 20, 2), -0.12, 0.12, 53)
    tpd = TPSSpatialTransformer(output_image_size=(16, 64), num_control_points=20, margins=(0.05, 0.05)).to(dev)
    out, src = tpd(x.to(dev), ctrl.to(dev))
    assert_close(src[:, ::7], t(g["tps_src"]), 5e-5, 5e-5, "source coordinates")
    assert_close(out, t(g["tps_out"]), 1e-3, 1e-3, "warped image")
    # a different input size than the output grid (tsrn.py feeds 32x64 into a 16x64 grid) and the identity warp
    x2 = synth.uniform("stn_x2", (3, 4, 32, 64), 0, 1, 54)
    ident = tp.target_control_points[None].repeat(3, 1, 1)
    o2, s2 = tpd(x2.to(dev), ident.to(dev))
    r2, rs2 = ostn.tps_forward(tp.inverse_kernel, tp.target_coordinate_repr, (16, 64), x2, ident)
    assert_close(s2, rs2, 5e-5, 5e-5, "identity source coordinates")
    assert_close(o2, r2, 1e-3, 1e-3, "identity warp, 32x64 -> 16x64")
    with pytest.raises(AssertionError):
        tpd(x.to(dev), ctrl[:, :19].to(dev))


def test_stn_head_rejects_the_tsrn_input_size_like_the_reference(dev):
    g = load_golden("stn_fwd")
    m = _head(dev, sd_from_manifest(g["manifest"], 52)).train()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 4, 32, 64, device=dev))          # quirk Q10: 1024 features into Linear(512, 512)

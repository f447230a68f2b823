"""Pin the CMM / DistillModule / ImageLoss / PSNR / SSIM / toMask oracle (CPU only)."""
import numpy as np
import pytest
import torch

from dpmn_amd.utils import synth
from oracle import cmm as ocmm
from helpers import load_golden, sd_from_manifest, checksum, t, assert_close


@pytest.mark.parametrize("cnum", [8, 64])
def test_cmm_matches_reference(cnum):
    g = load_golden("cmm_cnum%d" % cnum)
    sd = sd_from_manifest(g["manifest"], 31)
    assert abs(checksum(sd) - float(g["checksum"])) < 1e-6 * max(1.0, abs(float(g["checksum"])))
    x1 = synth.uniform("cmm_x1", (2, 3, 32, 128), 0, 1, 7)
    x2 = synth.uniform("cmm_x2", (2, 3, 32, 128), 0, 1, 7)
    assert_close(ocmm.cmm_forward(sd, x1, x2, False), t(g["out_eval"]), 2e-5, 1e-5, "cmm eval")
    assert_close(ocmm.cmm_forward(sd, x1, x2, True), t(g["out_train"]), 5e-5, 1e-5, "cmm train")


def test_distill_matches_reference():
    g = load_golden("distill")
    sd = sd_from_manifest(g["manifest"], 32)
    xd = synth.uniform("dist_deep", (2, 3, 32, 128), 0, 1, 8)
    xs = synth.uniform("dist_shallow", (2, 3, 32, 128), 0, 1, 8)
    for mode, tr in (("train", True), ("eval", False)):
        loss, feat = ocmm.distill_forward(sd, xd, xs, tr)
        assert_close(loss, t(g["loss_" + mode]), 1e-6, 1e-5, "distill loss " + mode)
        assert_close(feat, t(g["feat_" + mode]), 1e-5, 1e-5, "distill feat " + mode)


def test_loss_and_metrics_match_reference():
    g = load_golden("loss_metrics")
    a = synth.uniform("loss_a", (4, 3, 32, 128), 0, 1, 9)
    b = synth.uniform("loss_b", (4, 4, 32, 128), 0, 1, 9)
    b3 = 0.7 * a + 0.3 * b[:, :3]
    b4 = torch.cat([b3, b[:, 3:]], 1)
    assert_close(ocmm.image_loss(a, b3, True), t(g["loss_grad"]), 1e-7, 1e-5, "loss grad")
    assert_close(ocmm.image_loss(a, b3, False), t(g["loss_nograd"]), 1e-7, 1e-5, "loss nograd")
    assert_close(ocmm.gradient_map(a)[:1], t(g["gradmap"]), 1e-6, 1e-5, "gradient map")
    assert_close(ocmm.psnr(a, b4), t(g["psnr"]), 1e-4, 0, "psnr")
    assert_close(ocmm.ssim(a, b4), t(g["ssim"]), 1e-6, 0, "ssim")


def test_to_mask_matches_pil():
    """util.py:27-35 with PIL's own RGB->L conversion (torchvision is absent: the float->uint8
    step is restated as mul(255).byte(), quirk Q13)."""
    from PIL import Image
    img = synth.uniform("mask_img", (3, 3, 32, 128), -0.2, 1.2, 10)  # includes out-of-range values
    got = ocmm.to_mask(img)
    for i in range(img.shape[0]):
        u8 = (img[i] * 255).to(torch.int64).remainder(256).to(torch.uint8).permute(1, 2, 0).numpy()
        L = np.array(Image.fromarray(u8, "RGB").convert("L"))
        thr = L.mean()
        ref = np.where(L > thr, 0, 255).astype(np.float32) / 255.0
        assert np.array_equal(got[i, 0].numpy(), ref)
        assert torch.equal(got[i, 0], got[i, 1]) and torch.equal(got[i, 0], got[i, 2])

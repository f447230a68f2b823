"""GPU parity of the in-loop text-prior path: batched VisionLAN eval forward vs the imported
reference's outputs (tests/golden/visionlan.npz) and the oracle, the decode, the resize, the glyph-atlas composer (vs its
specification oracle/visionlan.py), and the whole pipeline driving TextSR.refine."""
import pytest
import torch

from dpmn_amd.utils import synth
from helpers import load_golden, sd_from_manifest, t, assert_close, record, max_abs_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    from dpmn_amd.model.visionlan import VisionLAN
    g = load_golden("visionlan")
    m = VisionLAN()
    assert [str(r).split("|")[0] for r in g["manifest"]] == list(m.state_dict().keys())
    sd = sd_from_manifest(g["manifest"], 61)
    m.load_state_dict(sd, strict=False)      # integer buffers (num_batches_tracked) are not part of the synthetic fill
    return m.to(dev).eval(), sd, g


def test_visionlan_forward_vs_reference_golden_and_oracle(dev, model):
    from oracle import visionlan as ov
    m, sd, g = model
    x = synth.uniform("vl_img", (3, 3, 64, 256), 0, 1, 62)
    from dpmn_amd import ops
    feat = m.features(ops.nchw_to_nhwc(x.to(dev), 4))
    with torch.no_grad():
        ref_feat = ov.backbone(sd, x)
    record("visionlan", "ResNet45 features max|err|", max_abs_err(feat.permute(0, 3, 1, 2), ref_feat), 9e-5)
    assert_close(feat.permute(0, 3, 1, 2), ref_feat, 9e-5, 9e-5, "ResNet45 feature map vs oracle")
    assert_close(feat.permute(0, 3, 1, 2)[:, ::16, :, ::4], t(g["feat_sample"]), 2e-4, 2e-4, "ResNet45 vs reference golden")
    lg = m.logits_from_features(feat)
    record("visionlan", "logits max|err| vs reference", max_abs_err(lg, t(g["logits"])), 1.5e-5)
    assert_close(lg, t(g["logits"]), 1.5e-5, 1.5e-5, "per-step logits vs reference golden")
    rows, length = m(x.to(dev), None, '', False)          # the reference's own call signature and return pair
    assert [int(v) for v in length.tolist()] == [int(v) for v in g["out_length"]]
    assert_close(rows, t(g["output"]), 5e-4, 5e-4, "(output, out_length) vs reference")
    from dpmn_amd.model.visionlan import decode_strings
    _, cls, ln = m.recognise(ops.nchw_to_nhwc(x.to(dev), 4))
    assert decode_strings(cls, ln) == [str(s) for s in g["texts"]]


def test_visionlan_batch_48_rows_match_small_batches(dev, model):
    """batched inference = per-image inference (the reference loops images at batch 1): rows of a B = 48 call equal B = 3 calls."""
    m, _, _ = model
    from dpmn_amd import ops
    x = synth.uniform("vl_img48", (48, 3, 32, 128), 0, 1, 65).to(dev)
    lg, cls, ln = m.recognise(x)
    lg3, cls3, ln3 = m.recognise(x[21:24].contiguous())
    assert_close(lg[21:24], lg3, 2e-5, 2e-5, "B=48 vs B=3 logits")
    assert torch.equal(cls[21:24], cls3) and torch.equal(ln[21:24], ln3)


def test_decode_and_resize_and_compose_vs_oracle(dev):
    from dpmn_amd import ops
    from oracle import visionlan as ov
    g = load_golden("visionlan")
    crafted = synth.uniform("vl_crafted", (4, 26, 37), -1, 1, 64)
    crafted[:, :, 0] -= 3.0
    crafted[0, 6, 0] = 5.0; crafted[0, 9, 0] = 5.0
    crafted[1, 0, 0] = 5.0
    crafted[3, 24, 0] = 5.0
    cls, ln = ops.vl_decode(crafted.to(dev).contiguous(), 25)
    rc, rl, texts = ov.decode(crafted)
    assert ln.tolist() == rl.tolist() == [int(v) for v in g["decode_length"]]
    from dpmn_amd.model.visionlan import decode_strings
    assert decode_strings(cls, ln) == texts == [str(s) for s in g["decode_texts"]]
    for b in range(4):
        assert cls[b, :int(ln[b])].tolist() == rc[b, :int(rl[b])].tolist()
    img = synth.uniform("vl_small", (5, 4, 32, 128), -0.1, 1.1, 66)
    got = ops.vl_resize(img.to(dev))
    ref = ov.resize_for_visionlan(img[:, :3])
    assert_close(got[..., :3].permute(0, 3, 1, 2), ref, 1e-6, 0, "parse_visionlan_data resize")
    assert float(got[..., 3].abs().max()) == 0.0
    # composer vs its specification, with ragged advances and an empty string
    GH, GW = 16, 12
    atlas = torch.floor(synth.uniform("atlas", (2, 37, GH, GW), 0, 256, 63))
    adv = torch.full((2, 37), GW, dtype=torch.int32)
    adv[:, 5] = 7; adv[1, 9] = 3
    c = torch.zeros(4, 25, dtype=torch.int32)
    c[0, :4] = torch.tensor([3, 5, 0, 9]); c[2, :2] = torch.tensor([7, 0]); c[3] = torch.arange(25, dtype=torch.int32) % 36 + 1
    lens = torch.tensor([3, 1, 2, 25], dtype=torch.int32)
    got = ops.text_prior_compose(c.to(dev), lens.to(dev), atlas.to(dev), adv.to(dev), 32, 128)
    ref = ov.compose_text_prior(c.long(), lens.long(), atlas, adv.long())
    d = (got.cpu() - ref).abs()
    assert float(d.max()) <= 1.0 and float((d > 0).float().mean()) < 2e-3, "composer vs specification (rounding ties only)"


def test_text_prior_pipeline_drives_refine(dev, model):
    """TextSR.refine with the recogniser-driven prior instead of synthetic priors: cfg0 stack, priors are uint8-valued images
    of the right shape, the result equals refine() fed the same priors explicitly."""
    from dpmn_amd import workload
    from dpmn_amd.interfaces.text_prior import VisionLANTextPrior
    m, _, _ = model
    sr, models, psn, inp = workload.build("cfg0")
    fn = VisionLANTextPrior([m], dev)
    seen = []
    wrapped = lambda cascade, k: seen.append(fn(cascade, k)) or seen[-1]
    out = sr.refine(models, psn, inp["images_lr"], None, text_prior_fn=wrapped)
    assert len(seen) == 1 and seen[0].shape == (4, 2, 32, 128)
    assert torch.equal(seen[0], torch.floor(seen[0])) and 0 <= float(seen[0].min()) and float(seen[0].max()) <= 255
    assert len(fn.strings()) == 4
    assert torch.equal(out, sr.refine(models, psn, inp["images_lr"], None, text_priors=seen))

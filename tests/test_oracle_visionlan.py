"""oracle/visionlan.py (eval-mode VisionLAN restatement + decode) against the imported reference's outputs
(tests/golden/visionlan.npz, tools/gen_golden.py gen_visionlan).  CPU only."""
import torch

from dpmn_amd.utils import synth
from helpers import load_golden, sd_from_manifest, t, assert_close, checksum


def _inputs():
    g = load_golden("visionlan")
    sd = sd_from_manifest(g["manifest"], 61)
    assert abs(checksum(sd) - float(g["checksum"])) < 1e-3 * abs(float(g["checksum"])) + 1e-3
    return g, sd, synth.uniform("vl_img", (3, 3, 64, 256), 0, 1, 62)


def test_visionlan_oracle_vs_reference_golden():
    from oracle import visionlan as ov
    g, sd, x = _inputs()
    assert ov.DICT36 == str(g["dict36"])
    assert_close(ov.pos_table()[::5, ::7], t(g["pos_table_sample"]), 1e-6, 0, "sinusoid table of the constructor")
    with torch.no_grad():
        feat = ov.backbone(sd, x)
        assert_close(feat[:, ::16, :, ::4], t(g["feat_sample"]), 2e-4, 2e-4, "ResNet45 feature map")
        lg = ov.logits(sd, x)
    assert_close(lg, t(g["logits"]), 2e-4, 2e-4, "per-step logits")
    cls, lengths, texts = ov.decode(lg)
    assert lengths.tolist() == [int(v) for v in g["out_length"]]
    assert texts == [str(s) for s in g["texts"]]
    rows = torch.cat([lg[b, :int(lengths[b])] for b in range(lg.shape[0])])
    assert_close(rows, t(g["output"]), 2e-4, 2e-4, "flattened (output, out_length) pair")
    # decode loop on crafted logits (EOS mid-string, at step 0, at the last step, never)
    crafted = synth.uniform("vl_crafted", (4, 26, 37), -1, 1, 64)
    crafted[:, :, 0] -= 3.0
    crafted[0, 6, 0] = 5.0; crafted[0, 9, 0] = 5.0
    crafted[1, 0, 0] = 5.0
    crafted[3, 24, 0] = 5.0
    cls, lengths, texts = ov.decode(crafted)
    assert lengths.tolist() == [int(v) for v in g["decode_length"]] == [7, 1, 25, 25]
    assert texts == [str(s) for s in g["decode_texts"]] and texts[1] == "" and len(texts[0]) == 6
    rows = torch.cat([crafted[b, :int(lengths[b])] for b in range(4)])
    assert torch.equal(rows, t(g["decode_output"]))


def test_text_prior_composer_properties():
    """The glyph-atlas composer is specified by the oracle (the reference's pygame renderer cannot run here): check its
    contract -- uint8-valued output, case channels, blank glyph for an empty string, stretch of a single glyph."""
    from oracle import visionlan as ov
    GH, GW = 16, 12
    atlas = torch.floor(synth.uniform("atlas", (2, 37, GH, GW), 0, 256, 63))
    adv = torch.full((2, 37), GW, dtype=torch.long)
    adv[:, 5] = 7
    cls = torch.tensor([[3, 5, 0, 9] + [0] * 21, [0] * 25, [7] + [0] * 24])
    lengths = torch.tensor([3, 1, 2])
    out = ov.compose_text_prior(cls, lengths, atlas, adv)
    assert out.shape == (3, 2, 32, 128) and torch.equal(out, torch.floor(out)) and out.min() >= 0 and out.max() <= 255
    blank = ov.compose_text_prior(torch.zeros(1, 25, dtype=torch.long), torch.tensor([25]), atlas, adv)
    assert torch.equal(out[1], blank[0])                         # empty string -> glyph 0
    one = atlas[:, 7]                                             # a single glyph stretched to 32x128: constant atlas -> constant image
    flat = ov.compose_text_prior(cls[2:], lengths[2:], torch.full_like(atlas, 200.0), adv)
    assert torch.equal(flat, torch.full_like(flat, 200.0))
    assert out[2].shape == (2, 32, 128) and float((out[2, 0] - out[2, 1]).abs().max()) > 0      # lower / upper atlases differ

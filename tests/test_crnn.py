"""The CRNN mirror that feeds TATT's label_vecs on real data (dpmn_amd/model/crnn.py) against the imported reference class
(tests/golden/crnn.npz, tools/gen_golden.py gen_crnn): state_dict layout, logits and the (B, 37, 1, 26) label vectors.  Stock
torch operators, so the check runs on the CPU."""
import torch

from dpmn_amd.model.crnn import CRNN
from dpmn_amd.utils import synth
from helpers import load_golden, sd_from_manifest, t, assert_close, checksum


def test_crnn_mirror_matches_reference_fixture():
    g = load_golden("crnn")
    m = CRNN(32, 1, 37, 256).eval()
    assert [str(r) for r in g["manifest"]] == ["%s|%s|%s" % (k, ",".join(map(str, v.shape)), str(v.dtype).replace("torch.", ""))
                                               for k, v in m.state_dict().items()], "state_dict layout differs from the reference CRNN"
    sd = m.state_dict()
    synth.synth_fill_(sd, seed=71)
    m.load_state_dict(sd)
    assert abs(checksum(sd) - float(g["checksum"])) < 1e-6 * abs(float(g["checksum"]))
    imgs = synth.uniform("crnn_lr", (3, 3, 16, 64), 0, 1, 72)
    with torch.no_grad():
        logits = m(m.parse_crnn_data(imgs))
    assert_close(logits, t(g["logits"]), 1e-5, 1e-5, "CRNN logits vs the reference")
    lv = m.label_vecs(imgs)
    assert lv.shape == (3, 37, 1, 26)
    assert_close(lv, t(g["label_vecs"]), 1e-6, 1e-5, "label_vecs (super_resolution.py:165-169)")

"""GPU half of the TextZoom collate (csrc/misc.hip k_collate_u8) against the fixture produced by the imported reference classes
(tests/golden/collate.npz): uint8 pixels in, ToTensor + mask channel out, bit for bit."""
import pytest
import torch

from dpmn_amd.dataset import textzoom as tz
from helpers import load_golden
from test_dataset import _fake_env

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mask", [True, False])
def test_gpu_collate_equals_reference_fixture(mask):
    dev = torch.device("cuda:0")
    g = load_golden("collate")
    env, _, _ = _fake_env()
    ds = tz.lmdbDataset_real(env=env, voc_type='all')
    out = tz.alignCollate_realWTLAMask(imgH=32, imgW=128, down_sample_scale=2, mask=mask, gpu_finish=True)([ds[i] for i in range(5)])
    (hr, lr, lv, strs), = list(tz.sr_batches([out], dev, mask))
    tag = "mask" if mask else "nomask"
    assert hr.is_cuda and hr.shape == ((5, 4, 32, 128) if mask else (5, 3, 32, 128))
    assert torch.equal(hr.cpu(), torch.from_numpy(g["hr_" + tag])), "HR batch differs from the reference collate"
    assert torch.equal(lr.cpu(), torch.from_numpy(g["lr_" + tag])), "LR batch differs from the reference collate"
    assert strs == [str(s) for s in g["label_strs"]] and lv is None


def test_tatt_trains_on_reader_batches_with_crnn_label_vecs(tmp_path):
    """--arch tatt on real-data batches (died on the zero text_emb): dataset reader -> GPU collate -> label_vecs from
    the frozen CRNN (<resume>/recognizer_best_crnn.pth, super_resolution.py:92, 165-169) -> two training steps and an eval."""
    from dpmn_amd import workload
    from dpmn_amd.interfaces.super_resolution import TextSR
    from dpmn_amd.model.crnn import CRNN
    from dpmn_amd.model.tatt import TSRN_TL_TRANS
    from dpmn_amd.utils import synth
    dev = torch.device("cuda:0")
    B = 4
    args = workload.make_args("tatt", 1, 1, B)
    args.resume = str(tmp_path)
    cfg = workload.make_config(B)
    cfg.TRAIN.ckpt_dir = str(tmp_path / "ckpt")
    cfg.TRAIN.saveInterval, cfg.TRAIN.displayInterval = 100, 100
    psn = TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32)
    sd = psn.state_dict(); synth.synth_fill_(sd, 41)
    torch.save({'state_dict_G': sd}, tmp_path / "model_tatt.pth")
    crnn = CRNN(32, 1, 37, 256)
    sdc = crnn.state_dict(); synth.synth_fill_(sdc, 71)
    torch.save(sdc, tmp_path / "recognizer_best_crnn.pth")
    env, _, _ = _fake_env()
    ds = tz.lmdbDataset_real(env=env, voc_type='all')
    col = tz.alignCollate_realWTLAMask(imgH=32, imgW=128, down_sample_scale=2, mask=True, gpu_finish=True)
    collated = [col([ds[i] for i in (0, 1, 3, 4)]), col([ds[i] for i in (4, 3, 1, 0)])]
    sr = TextSR(cfg, args)
    sr.vis_dir = str(tmp_path / "vis")
    seen = []
    orig = sr.label_vecs_from_crnn
    sr.label_vecs_from_crnn = lambda lr: seen.append(orig(lr)) or seen[-1]
    models, distill = sr.train(lambda epoch: tz.sr_batches(collated, dev, True), epochs=1)
    assert len(seen) == 2 and seen[0].shape == (B, 37, 1, 26) and torch.isfinite(seen[0]).all()
    assert float((seen[0].sum(1) - 1).abs().max()) < 1e-5, "label_vecs are per-position class distributions"
    _, psn_m = sr.build_models()
    md = sr.eval(models, tz.sr_batches(collated[:1], dev, True), model_psn=psn_m, text_prior_fn=sr.synthetic_text_prior())
    assert len(seen) == 3 and md["psnr_avg"] == md["psnr_avg"] and 0.0 <= md["ssim_avg"] <= 1.0

"""Data path (dpmn_amd/dataset/textzoom.py) against direct PIL calls -- the library the reference itself
calls in dataset.py:1266-1319 -- over an in-memory stand-in for the LMDB environment (lmdb is not installed here)."""
import io

import numpy as np
import torch
from PIL import Image

from dpmn_amd.dataset import textzoom as tz


class _Txn:
    def __init__(self, d):
        self.d = d

    def get(self, k):
        return self.d.get(k)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Env:
    def __init__(self, d):
        self.d = d

    def begin(self, write=False):
        return _Txn(self.d)


def _png(arr):
    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="PNG")
    return b.getvalue()


def _fake_env(n=5):
    rng = np.random.RandomState(7)
    d = {b'num-samples': str(n).encode()}
    words = ["Hello", "a", "", "SuperResolution-2023!", "abcdefghijklmnopqrstuvwxyz0123"]
    raw = []
    for i in range(1, n + 1):
        hr = rng.randint(0, 256, (40 + 3 * i, 150 + 7 * i, 3)).astype(np.uint8)
        lr = rng.randint(0, 256, (18 + i, 70 + 3 * i, 3)).astype(np.uint8)
        d[b'image_hr-%09d' % i], d[b'image_lr-%09d' % i] = _png(hr), _png(lr)
        d[b'label-%09d' % i] = words[i - 1].encode()
        raw.append((hr, lr))
    return _Env(d), raw, words


def test_dataset_reader_and_resize_normalize_match_pil():
    env, raw, words = _fake_env()
    ds = tz.lmdbDataset_real(env=env, voc_type='all', max_len=100, test=False)
    assert len(ds) == 5
    hr, lr, _, _, s = ds[0]
    assert hr.size == (157, 43) and lr.size == (73, 19) and s == "Hello"
    assert np.array_equal(np.asarray(hr), raw[0][0])
    assert ds[3][4] == "SuperResolution-2023!" and tz.lmdbDataset_real(env=env, voc_type='lower')[3][4] == "superresolution2023"
    t = tz.resizeNormalize((128, 32), mask=True)(hr)
    ref_img = Image.fromarray(raw[0][0]).resize((128, 32), Image.BICUBIC)
    ref = torch.from_numpy(np.asarray(ref_img).transpose(2, 0, 1).copy()).float() / 255
    assert t.shape == (4, 32, 128) and torch.equal(t[:3], ref)
    gray = np.asarray(ref_img.convert('L'))
    assert torch.equal(t[3], torch.from_numpy(np.where(gray > gray.mean(), 0, 255).astype(np.uint8)).float() / 255)
    assert set(t[3].unique().tolist()) <= {0.0, 1.0}


def test_collate_tuple_layout_and_label_vectors():
    env, raw, words = _fake_env()
    ds = tz.lmdbDataset_real(env=env, voc_type='all')
    col = tz.alignCollate_realWTLAMask(imgH=32, imgW=128, down_sample_scale=2, mask=True)
    out = col([ds[i] for i in range(5)])
    assert len(out) == 9 and out[1] is None and out[3] is None and out[4] is None
    hr, lr, strs, vecs, wm, wt = out[0], out[2], out[5], out[6], out[7], out[8]
    assert hr.shape == (5, 4, 32, 128) and lr.shape == (5, 4, 16, 64) and vecs.shape == (5, 37, 1, 26)
    assert list(strs) == [tz.str_filt(w, 'all') for w in words]
    # "hello": 5 letters spread over 26 slots with int(21 / 4) = 5 blanks in each of the 4 gaps -> 25 one-hot positions
    v = vecs[0, :, 0]                                   # (37 classes, 26 positions)
    assert float(v.sum()) == 25.0 and int(v[:, 0].argmax()) == col.a2d['h'] and int(v[:, 6].argmax()) == col.a2d['e']
    assert int(v[0].sum()) == 20                        # the blanks are class 0 ('-')
    assert wt.tolist() == [1, 1, 0, 1, 1] and float(vecs[2, 0, 0, 0]) == 1.0 and float(vecs[2].sum()) == 1.0
    assert float(vecs[4].sum()) == 26.0                 # 30 characters truncated to 26
    batches = list(tz.sr_batches([out]))
    assert batches[0][0] is hr and batches[0][1] is lr and batches[0][2] is None and batches[0][3] == list(strs)


def _fixture_batch():
    from helpers import load_golden
    g = load_golden("collate")
    env, raw, words = _fake_env()
    ds = tz.lmdbDataset_real(env=env, voc_type='all')
    return g, [ds[i] for i in range(5)]


def test_collate_equals_the_reference_class_fixture():
    """tests/golden/collate.npz = outputs of the IMPORTED reference classes (tools/gen_golden.py gen_collate: resizeNormalize,
    alignCollate_realWTLAMask.__call__, str_filt of dataset/dataset.py and utils/util.py) on the same five synthetic images."""
    g, batch = _fixture_batch()
    for mask, tag in ((True, "mask"), (False, "nomask")):
        out = tz.alignCollate_realWTLAMask(imgH=32, imgW=128, down_sample_scale=2, mask=mask)(batch)
        assert torch.equal(out[0], torch.from_numpy(g["hr_" + tag])) and torch.equal(out[2], torch.from_numpy(g["lr_" + tag]))
        if mask:
            assert list(out[5]) == [str(s) for s in g["label_strs"]]
            assert torch.equal(out[6], torch.from_numpy(g["label_vecs"]))
            assert out[7].tolist() == g["weighted_masks"].tolist() and out[8].tolist() == g["weighted_tics"].tolist()
    for voc in ("lower", "upper", "all", "digit"):
        assert [tz.str_filt(str(w), voc) for w in g["str_filt_in"]] == [str(s) for s in g["str_filt_" + voc]], voc


def test_gpu_finish_collate_returns_the_resized_uint8_pixels():
    g, batch = _fixture_batch()
    out = tz.alignCollate_realWTLAMask(imgH=32, imgW=128, down_sample_scale=2, mask=True, gpu_finish=True)(batch)
    assert out[0].dtype == torch.uint8 and out[0].shape == (5, 32, 128, 3) and out[2].shape == (5, 16, 64, 3)
    # the uint8 pixels are what ToTensor divides by 255 in the reference
    assert torch.equal(out[0].permute(0, 3, 1, 2).float() / 255, torch.from_numpy(g["hr_nomask"]))
    assert torch.equal(out[6], torch.from_numpy(g["label_vecs"]))

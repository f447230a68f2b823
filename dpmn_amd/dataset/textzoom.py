"""TextZoom data path of the SR trainer: the parts of dataset/dataset.py that
interfaces/base.py:85-125 wires into the training / evaluation loaders, restricted to what the SR path consumes.

  lmdbDataset_real          dataset.py:565-686   LMDB reader: keys image_hr-%09d / image_lr-%09d / label-%09d / num-samples
  resizeNormalize           dataset.py:1266-1319 PIL bicubic resize -> ToTensor -> optional mask channel (gray > mean ? 0 : 255)
  alignCollate_realWTLAMask dataset.py:1966-2076 batch of (HR, LR, label) -> the tuple TextSR.train unpacks (super_resolution.py:142)

Host code by nature (JPEG / PNG decode and PIL resampling are what the reference does too); PIL is the same library the
reference calls, so `resizeNormalize` is not a restatement of an algorithm but the same calls.  Left out, with the positions
kept in the collated tuple: the YUV copies (cv2.cvtColor, read by nothing on the SR path), imgaug augmenters (constructed but
never applied by these classes), cutblur / manmade degradation (off in interfaces/base.py).  `lmdb` itself is imported lazily:
it is not installed in the build image, so the reader is exercised in tests/ through an injected environment object with the
same `begin().get(key)` protocol, over images encoded the way TextZoom stores them.
"""
import io

import numpy as np
import torch

from ..utils.util import str_filt


def buf2PIL(txn, key, mode='RGB'):
    """dataset.py:54-60."""
    from PIL import Image
    buf = txn.get(key)
    if buf is None:
        raise IOError("missing key %r" % key)
    return Image.open(io.BytesIO(buf)).convert(mode)


def _to_tensor(img):
    """torchvision.transforms.ToTensor for a PIL image of mode RGB / L: uint8 HWC -> float CHW / 255."""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255.0)


class lmdbDataset_real(torch.utils.data.Dataset):
    def __init__(self, root=None, voc_type='upper', max_len=100, test=False, cutblur=False, manmade_degrade=False, rotate=None,
                 env=None):
        super().__init__()
        if cutblur or manmade_degrade:
            raise NotImplementedError("dpmn_amd lmdbDataset_real: cutblur / manmade_degrade are off on the SR path (base.py:85-125)")
        if env is None:
            try:
                import lmdb
            except ImportError as e:
                raise RuntimeError("dpmn_amd: reading TextZoom needs the `lmdb` package (not installed here); "
                                   "main.py falls back to synthetic batches with --synthetic_steps") from e
            env = lmdb.open(root, max_readers=1, readonly=True, lock=False, readahead=False, meminit=False)
        self.env = env
        with self.env.begin(write=False) as txn:
            self.nSamples = int(txn.get(b'num-samples'))
        self.voc_type, self.max_len, self.test = voc_type, max_len, test

    def __len__(self):
        return self.nSamples

    def __getitem__(self, index):
        assert index <= len(self), 'index range error'
        index += 1
        with self.env.begin(write=False) as txn:
            try:
                img_HR = buf2PIL(txn, b'image_hr-%09d' % index, 'RGB')
                img_lr = buf2PIL(txn, b'image_lr-%09d' % index, 'RGB')
                word = txn.get(b'label-%09d' % index)
                word = " " if word is None else str(word.decode())
            except IOError:
                return self[index % len(self)]          # dataset.py:680-681: skip to the next sample (index is already +1)
        return img_HR, img_lr, None, None, str_filt(word, self.voc_type)


class resizeNormalize(object):
    def __init__(self, size, mask=False, interpolation=None):
        from PIL import Image
        self.size, self.mask = size, mask
        self.interpolation = Image.BICUBIC if interpolation is None else interpolation

    def __call__(self, img):
        img = img.resize(self.size, self.interpolation)
        t = _to_tensor(img)
        if self.mask:
            m = img.convert('L')
            thres = np.array(m).mean()
            m = m.point(lambda x: 0 if x > thres else 255)
            t = torch.cat((t, _to_tensor(m)), 0)
        return t


class resizeU8(object):
    """Host half of the GPU collate: the same PIL bicubic resize as resizeNormalize, returned as the raw uint8 HWC pixels --
    ToTensor and the mask channel run on the GPU (csrc/misc.hip k_collate_u8 through finish_on_gpu)."""

    def __init__(self, size, interpolation=None):
        from PIL import Image
        self.size = size
        self.interpolation = Image.BICUBIC if interpolation is None else interpolation

    def __call__(self, img):
        return torch.from_numpy(np.asarray(img.resize(self.size, self.interpolation), dtype=np.uint8).copy())


def finish_on_gpu(images_u8, mask, device):
    """(B, H, W, 3) uint8 from a gpu_finish collate -> (B, 3 + mask, H, W) float on `device` (one 3-byte-per-pixel upload and one
    kernel per batch instead of B x (ToTensor + convert('L') + point + cat) on the host and a 16-byte-per-pixel upload)."""
    from .. import ops
    return ops.collate_u8(images_u8.to(device, non_blocking=True), mask)


class alignCollate_realWTLAMask(object):
    """collate_fn of the training loader (base.py:99-102); returns the 9-tuple of dataset.py:2076 with None in the positions of
    the YUV copies and the pseudo-LR batch.  gpu_finish=True (ours): positions 0 and 2 hold the resized uint8 (B, H, W, 3) pixels
    and `sr_batches(loader, device)` finishes them on the GPU -- same values, bit for bit (tests/test_gpu_dataset.py)."""

    def __init__(self, imgH=64, imgW=256, down_sample_scale=4, keep_ratio=False, min_ratio=1, mask=False, alphabet=53, train=True,
                 y_domain=False, gpu_finish=False):
        self.imgH, self.imgW, self.down_sample_scale, self.mask = imgH, imgW, down_sample_scale, mask
        self.gpu_finish = gpu_finish
        self.alphabet = "0123456789abcdefghijklmnopqrstuvwxyz"
        self.d2a = "-" + self.alphabet
        self.alsize = len(self.d2a)
        self.a2d = {ch: i for i, ch in enumerate(self.d2a)}
        if gpu_finish:
            self.transform = resizeU8((imgW, imgH))
            self.transform2 = resizeU8((imgW // down_sample_scale, imgH // down_sample_scale))
        else:
            self.transform = resizeNormalize((imgW, imgH), mask)
            self.transform2 = resizeNormalize((imgW // down_sample_scale, imgH // down_sample_scale), mask)

    def __call__(self, batch):
        images_HR, images_lr, _, _, label_strs = zip(*batch)
        images_HR = torch.stack([self.transform(im) for im in images_HR], 0)
        images_lr = torch.stack([self.transform2(im) for im in images_lr], 0)
        max_len = 26
        label_batches, weighted_masks, weighted_tics = [], [], []
        for word in label_strs:
            word = word.lower()
            if 1 < len(word) < 26:                       # spread the characters over 26 slots (dataset.py:2019-2027)
                padding = int((26 - len(word)) / (len(word) - 1))
                word = word[0] + "".join("-" * padding + ch for ch in word[1:])
            elif len(word) >= 26:
                word = word[:26]
            label_list = [self.a2d[ch] for ch in word if ch in self.a2d]
            if len(label_list) <= 0:
                weighted_masks.append(0)
            else:
                weighted_masks.extend(label_list)
            labels = torch.tensor(label_list, dtype=torch.long)[:, None]
            if labels.shape[0] > 0:
                label_batches.append(torch.zeros((labels.shape[0], self.alsize)).scatter_(-1, labels, 1))
                weighted_tics.append(1)
            else:
                vec = torch.zeros((1, self.alsize))
                vec[0, 0] = 1.
                label_batches.append(vec)
                weighted_tics.append(0)
        label_rebatches = torch.zeros((len(label_strs), max_len, self.alsize))
        for idx, lb in enumerate(label_batches):
            label_rebatches[idx][:lb.shape[0]] = lb
        label_rebatches = label_rebatches.unsqueeze(1).float().permute(0, 3, 1, 2)
        return images_HR, None, images_lr, None, None, label_strs, label_rebatches, torch.tensor(weighted_masks).long(), torch.tensor(weighted_tics)


def sr_batches(loader, device=None, mask=None):
    """Adapter for TextSR.train / eval / test: (images_hr, images_lr, label_vecs, label_strs) per batch.  label_vecs is None: for
    --arch tatt the reference derives them from a CRNN on the LR image (super_resolution.py:165-169), not from the dataset.
    Batches of a gpu_finish collate (uint8 pixels) are finished on `device` here (default: the current GPU); mask=None takes the
    collate function's own setting."""
    if mask is None:
        mask = bool(getattr(getattr(loader, "collate_fn", None), "mask", True))
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    for data in loader:
        hr, lr = data[0], data[2]
        if hr.dtype == torch.uint8:
            hr, lr = finish_on_gpu(hr, mask, device), finish_on_gpu(lr, mask, device)
        yield hr, lr, None, list(data[5])

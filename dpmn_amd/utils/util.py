"""Mirror of the helpers of ``utils/util.py`` that sit on the SR path."""
import os
import random

import numpy as np
import torch

from .. import ops


def set_seed(seed):
    """utils/util.py:16-25."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    random.seed(seed)
    np.random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)


def toMask(img_tensor):
    """utils/util.py:27-35 for ONE image (3,H,W) -> (1,3,H,W); prefer ops.to_mask for batches."""
    return ops.to_mask(img_tensor[None])


def torch_rotate_img(torch_image_batches, arc_batches, rand_offs, off_range=0.2):
    """utils/util.py:37-58 -- same signature; one HIP kernel (affine_grid + bilinear grid_sample fused)."""
    return ops.rotate_img(torch_image_batches, arc_batches, rand_offs, off_range)


def str_filt(str_, voc_type):
    import string
    alpha_dict = {'digit': string.digits, 'lower': string.digits + string.ascii_lowercase,
                  'upper': string.digits + string.ascii_letters,
                  'all': string.digits + string.ascii_letters + string.punctuation}
    if voc_type == 'lower':
        str_ = str_.lower()
    return ''.join(ch for ch in str_ if ch in alpha_dict[voc_type])

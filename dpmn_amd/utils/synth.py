"""Deterministic synthetic weights and inputs (host code, no reference import).

The reference ships no weights and no data; every parity fixture and the
bench therefore use values that both sides can regenerate exactly:

* each tensor is drawn from its *own* CPU generator seeded by crc32(name) ^ seed, so the
  result does not depend on dict order or on which other tensors exist;
* only ``torch.rand`` is used (uniform integers -> float, no transcendental), which is
  bit-reproducible across CPU ISAs, unlike the vectorised Box-Muller behind ``randn``.

Input conventions follow SURVEY.md §8(d) (seed 2 = ``manualSeed`` of the reference yaml,
config/super_resolution.yaml:25).
"""
import math
import zlib

import torch

_SKIP = ("relative_position_index", "attn_mask", "num_batches_tracked")


def _gen(name, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def uniform(name, shape, lo, hi, seed=0):
    return torch.rand(tuple(shape), generator=_gen(name, seed), dtype=torch.float32) * (hi - lo) + lo


def synth_tensor(name, shape, seed=0):
    """Value rule by parameter name; keeps activations O(1) through deep stacks."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if "relative_position_bias_table" in name:
        return uniform(name, shape, -0.5, 0.5, seed)
    if leaf.startswith("weight_list"):
        return uniform(name, shape, 0.8, 1.2, seed)
    if leaf == "running_var":
        return uniform(name, shape, 0.5, 1.5, seed)
    if leaf == "running_mean":
        return uniform(name, shape, -0.2, 0.2, seed)
    if leaf in ("a_2",):
        return uniform(name, shape, 0.8, 1.2, seed)
    if leaf in ("b_2",):
        return uniform(name, shape, -0.1, 0.1, seed)
    if len(shape) <= 1:
        if leaf == "weight":
            # 1-D weights are LayerNorm/BatchNorm scales or PReLU slopes
            if shape == (1,):
                return uniform(name, shape, 0.15, 0.35, seed)
            return uniform(name, shape, 0.8, 1.2, seed)
        return uniform(name, shape, -0.1, 0.1, seed)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    if "init_factor" in name:  # nn.Embedding (rows, dim): unit-scale rows
        fan_in = 3
    a = math.sqrt(3.0 / fan_in)
    return uniform(name, shape, -a, a, seed)


def synth_fill_(state_dict, seed=0):
    """Overwrite every floating-point entry of ``state_dict`` (in place) by the name rule."""
    for k, v in state_dict.items():
        if any(s in k for s in _SKIP) or not torch.is_floating_point(v):
            continue
        if k.endswith(".pe") or k == "pe":
            continue
        v.copy_(synth_tensor(k, v.shape, seed))
    return state_dict


# ---------------------------------------------------------------- synthetic batch
def _with_mask_channel(img3):
    """ch3 = (gray < mean) binary mask, mirrors resizeNormalize's mask (dataset.py:1312-1317)."""
    gray = 0.299 * img3[:, 0] + 0.587 * img3[:, 1] + 0.114 * img3[:, 2]
    thr = gray.mean(dim=(1, 2), keepdim=True)
    mask = (gray < thr).float().unsqueeze(1)
    return torch.cat([img3, mask], 1)


def synth_batch(B, seed=2, h_lr=16, w_lr=64, scale=2):
    """Returns dict(images_lr (B,4,h,w), images_hr (B,4,2h,2w), text_prior (B,2,2h,2w) uint8-valued
    floats (quirk Q6), label_vecs (B,37,1,26) rows summing to 1)."""
    lr = uniform("images_lr", (B, 3, h_lr, w_lr), 0.0, 1.0, seed)
    hr = uniform("images_hr", (B, 3, h_lr * scale, w_lr * scale), 0.0, 1.0, seed)
    prior = torch.floor(uniform("text_prior", (B, 2, h_lr * scale, w_lr * scale), 0.0, 256.0, seed))
    lv = uniform("label_vecs", (B, 37, 1, 26), 0.0, 1.0, seed) ** 4
    lv = lv / lv.sum(1, keepdim=True)
    return {
        "images_lr": _with_mask_channel(lr),
        "images_hr": _with_mask_channel(hr),
        "text_prior": prior,
        "label_vecs": lv,
    }

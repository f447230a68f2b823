"""Shared test helpers: rebuild synthetic state dicts from a golden manifest."""
import os

import numpy as np
import torch

from dpmn_amd.utils import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def sd_from_manifest(man, seed):
    """manifest rows 'name|d0,d1|dtype' -> synthetic state dict (float entries only by the
    synth rule; integer/derived buffers are omitted, the oracle/HIP path recompute them)."""
    sd = {}
    for row in man:
        name, shape, dtype = str(row).split("|")
        shape = tuple(int(s) for s in shape.split(",")) if shape else ()
        if dtype.startswith("float"):
            sd[name] = torch.zeros(shape, dtype=torch.float32)
        else:
            sd[name] = torch.zeros(shape, dtype=torch.int64)
    synth.synth_fill_(sd, seed=seed)
    return sd


def checksum(sd):
    skip = ("relative_position_index", "attn_mask", "num_batches_tracked")
    return float(sum(v.double().sum().item() for k, v in sd.items()
                     if torch.is_floating_point(v) and not any(s in k for s in skip) and not k.endswith(".pe")))


def t(a):
    return torch.from_numpy(np.asarray(a))


def assert_close(a, b, atol, rtol=0.0, what=""):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), "%s max err %.3e (tol %.1e) at %d/%d elems, ref max %.3e" % (
        what, err.max().item(), atol, int(bad.sum()), a.numel(), b.abs().max().item())

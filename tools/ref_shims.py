"""Import-time shims so the read-only reference at /root/reference can be imported in the
build container (CPU only) to generate golden vectors.  Nothing here executes on the hot
path: the stubbed modules are only *imported* by the reference, never called on the path
(SURVEY.md §8c).  This file is test tooling; it never travels into the product path.
"""
import sys
import types
import torch

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    # timm.models.layers.{DropPath,to_2tuple,trunc_normal_} (pgrm.py:10) -- timm 0.6.5 not installed
    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            mask = x.new_empty(shape).bernoulli_(keep)
            return x * mask / keep

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    timm = _stub("timm")
    timm.models = _stub("timm.models")
    timm.models.layers = _stub("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple,
                               trunc_normal_=torch.nn.init.trunc_normal_)
    _stub("IPython", embed=lambda *a, **k: None)
    _stub("cv2")
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.transforms = _stub("torchvision.transforms")
    return REF_ROOT

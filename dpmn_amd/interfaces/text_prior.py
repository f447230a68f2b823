"""The in-loop text prior of branch 1 (interfaces/super_resolution.py:174-199) as ONE batched GPU pipeline:
cascade image -> parse_visionlan_data resize -> VisionLAN (eval) -> per-step arg-max + length -> string laid out from a glyph
atlas in lower and upper case -> x_q (B, 2, H, W), uint8-valued floats (quirk Q6).  The reference does this per image on the
host (recogniser at batch 1, pygame + cv2, D2H / H2D per image); see DESIGN.md for what is and is not pinned."""
import torch

from .. import ops
from ..model.visionlan import VisionLAN, decode_strings
from ..utils import glyphs


class VisionLANTextPrior:
    """``text_prior_fn`` for TextSR.refine / train_step: fn(cascade, k) -> (B, 2, H, W).
    recognizers: one VisionLAN per branch-1 stage (super_resolution.py:100-111) or a single shared one."""

    def __init__(self, recognizers, device, font_path=None, atlas=None, advance=None, glyph_h=32):
        self.recognizers = list(recognizers) if isinstance(recognizers, (list, tuple)) else [recognizers]
        for r in self.recognizers:
            r.eval()
        if atlas is None:
            atlas, advance = glyphs.build_atlas(font_path, glyph_h, device)
        self.atlas, self.advance = atlas.to(device).contiguous(), advance.to(device).contiguous()
        self.last = None        # (classes, lengths) of the most recent call, for logging / accuracy

    @torch.no_grad()
    def __call__(self, cascade, k):
        rec = self.recognizers[min(k, len(self.recognizers) - 1)]
        _, cls, length = rec.recognise(cascade[:, :3])
        self.last = (cls, length)
        return ops.text_prior_compose(cls, length, self.atlas, self.advance, cascade.shape[2], cascade.shape[3])

    def strings(self):
        return decode_strings(*self.last) if self.last is not None else []


def _load_visionlan(m, path, device):
    """VisionLAN_init's key handling (interfaces/base.py:452-471): DataParallel 'module.' prefixes dropped, only keys the
    model owns are taken."""
    sd = torch.load(path, map_location=device)
    sd = {(k[7:] if k.startswith("module.") else k).replace("features.module.", "features."): v for k, v in sd.items()}
    own = m.state_dict()
    own.update({k: v for k, v in sd.items() if k in own})
    m.load_state_dict(own)


def build_recognizers(n, device, path=None, allow_random=False):
    """The student recognisers of super_resolution.py:100-111: stage i is loaded from
    os.path.join(rec_path, 'recognizer_best_%d.pth' % i) when `path` is a directory (the reference's --rec_path), from the
    single file for every stage when `path` is a file (VisionLAN_init(path)).  No checkpoint: the reference cannot run at all
    (rec_path None -> os.path.join raises), so a randomly initialised recogniser silently driving the text prior is refused
    unless the caller asks for it (tests, benchmarks on synthetic weights)."""
    import os
    if not path and not allow_random:
        raise RuntimeError("dpmn_amd: --tpg visionlan needs --rec_path (a directory holding recognizer_best_{i}.pth per stage, or "
                           "one visionlan .pth file); pass --synthetic_prior to train / evaluate on seeded noise priors instead")
    recs = []
    for i in range(n):
        m = VisionLAN().to(device)
        if path:
            f = os.path.join(path, "recognizer_best_%d.pth" % i) if os.path.isdir(path) else path
            if not os.path.isfile(f):
                raise FileNotFoundError("dpmn_amd: recogniser checkpoint %s is missing" % f)
            print('load pre_trained VisionLAN model from %s' % f)
            _load_visionlan(m, f, device)
        recs.append(m.eval())
    return recs

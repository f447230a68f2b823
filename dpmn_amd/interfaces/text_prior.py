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


def build_recognizers(n, device, path=None):
    """VisionLAN_init (interfaces/base.py:452-471): n recognisers, optionally initialised from a checkpoint whose keys may carry
    the 'module.' prefix of nn.DataParallel."""
    recs = []
    for _ in range(n):
        m = VisionLAN().to(device)
        if path:
            sd = torch.load(path, map_location=device)
            sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
            own = m.state_dict()
            own.update({k: v for k, v in sd.items() if k in own})
            m.load_state_dict(own)
        recs.append(m.eval())
    return recs

"""Glyph atlas for the GPU text-prior composer (csrc/visionlan.hip k_text_prior), the replacement of the reference's
per-string pygame rasteriser utils/render_standard_text.py (pygame and cv2 are not available here; that file is GPL, nothing
of it is reused).  One bitmap per (case, class): class 0 = the blank the reference draws for an empty string ("\\t"), class c
-> DICT36[c - 1], rendered ONCE on the host with PIL's FreeType (``--font_path`` of the CLI, or PIL's built-in font),
anti-aliased, white-on-black alpha 0..255 like ``pixels_alpha`` (render_standard_text.py:22), all on one baseline.
Per string the GPU then only gathers: no host round trip per image."""
import torch

from ..model.visionlan import DICT36


def build_atlas(font_path=None, glyph_h=32, device=None):
    """Returns (atlas (2, 37, GH, GW) float32 in 0..255, advance (2, 37) int32): case 0 lower, case 1 upper."""
    from PIL import Image, ImageDraw, ImageFont
    size = int(glyph_h * 0.8)
    font = ImageFont.truetype(font_path, size) if font_path else ImageFont.load_default(size)
    ascent, descent = font.getmetrics()
    scale_h = ascent + descent
    chars = [[" "] + list(DICT36), [" "] + list(DICT36.upper())]
    cells, adv = [], []
    gw = 0
    for case in range(2):
        row, arow = [], []
        for ch in chars[case]:
            w = max(1, int(round(font.getlength(ch))))
            img = Image.new("L", (w, scale_h), 0)
            ImageDraw.Draw(img).text((0, 0), ch, fill=255, font=font)
            if scale_h != glyph_h:
                img = img.resize((max(1, int(round(w * glyph_h / scale_h))), glyph_h), Image.BILINEAR)
            t = torch.frombuffer(bytearray(img.tobytes()), dtype=torch.uint8).reshape(glyph_h, img.size[0]).float()
            row.append(t)
            arow.append(img.size[0])
            gw = max(gw, img.size[0])
        cells.append(row)
        adv.append(arow)
    atlas = torch.zeros(2, 37, glyph_h, gw)
    for case in range(2):
        for c, t in enumerate(cells[case]):
            atlas[case, c, :, :t.shape[1]] = t
    advance = torch.tensor(adv, dtype=torch.int32)
    if device is not None:
        atlas, advance = atlas.to(device), advance.to(device)
    return atlas.contiguous(), advance.contiguous()

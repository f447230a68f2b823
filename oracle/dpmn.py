"""ORACLE (test infrastructure only) -- CPU restatement of the DPMN eval forward stack
(interfaces/super_resolution.py:370-449): frozen PSN -> branch-1 PGRMs (text prior) -> branch-2 PGRMs
(mask prior from toMask) -> CMM -> alpha blend.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file.  Pinned by tests/golden/stack_cfg0.npz (reference modules driven
by a restatement of those lines in tools/gen_golden.py; toMask itself is pinned against PIL).
"""
from . import cmm as ocmm
from . import pgrm as opgrm
from . import tsrn as otsrn


def refine(sd_psn, sd_pgrms, sd_cmm, arch, b1, b2, images_lr, label_vecs, text_priors, alpha=0.5, return_all=False,
           windows=(2, 4, 8)):
    if arch == "tatt":
        psn, _ = otsrn.tatt_forward(sd_psn, images_lr, label_vecs)
    elif arch == "tbsrn":
        psn = otsrn.tbsrn_forward(sd_psn, images_lr)
    elif arch == "tpgsr":
        psn = otsrn.tsrn_tl_forward(sd_psn, images_lr, label_vecs)
    else:
        psn = otsrn.tsrn_forward(sd_psn, images_lr)
    cascade, br1 = psn, []
    for k in range(b1):
        sr = opgrm.pgrm_forward(sd_pgrms[k], text_priors[k], cascade[:, :3], br1[:k], windows=windows)
        br1.append(sr)
        cascade = sr
    cascade, br2 = psn, []
    for k in range(b1, b1 + b2):
        sr = opgrm.pgrm_forward(sd_pgrms[k], ocmm.to_mask(cascade[:, :3]), cascade[:, :3], br2[:(k - b2)], windows=windows)
        br2.append(sr)
        cascade = sr
    fused = ocmm.cmm_forward(sd_cmm, br1[-1], br2[-1], False)
    out = alpha * fused + (1 - alpha) * psn[:, :3]
    if return_all:
        return out, dict(psn=psn, branch1=br1, branch2=br2, cmm=fused)
    return out


def train_loss(sd_psn, sd_pgrms, sd_distill, sd_cmm, arch, b1, b2, images_lr, images_hr, label_vecs, text_priors, windows=(2, 4, 8)):
    """The loss of one optimisation step (interfaces/super_resolution.py:140-262), differentiable by torch autograd through
    the restated modules: frozen PSN (no_grad, :56-59) -> both PGRM cascades with ImageLoss x 100 on every cascade image
    (:201-203, :232-234) -> the DistillModule chains walking each branch backwards (:237-252) -> CMM + its ImageLoss
    (:254-258) -> sum / (b1 + b2 + 1) (:262).  The state dicts hold the leaves (requires_grad set by the caller); the
    DistillModules are ordered [branch-1 chain (b1 - 1), branch-2 chain (b2 - 1)].  Pinned through
    tests/golden/step_tsrn_2p2.npz (tests/test_oracle_grads.py)."""
    import torch
    with torch.no_grad():
        if arch == "tatt":
            psn, _ = otsrn.tatt_forward(sd_psn, images_lr, label_vecs)
        elif arch == "tbsrn":
            psn = otsrn.tbsrn_forward(sd_psn, images_lr)
        else:
            psn = otsrn.tsrn_forward(sd_psn, images_lr)
    hr3 = images_hr[:, :3]
    tot, casc, l1, l2 = 0, psn, [], []
    for k in range(b1):
        o = opgrm.pgrm_forward(sd_pgrms[k], text_priors[k], casc[:, :3], l1[:k], windows=windows)
        l1.append(o)
        casc = o
        tot = tot + ocmm.image_loss(o, hr3, True) * 100
    casc = psn
    for k in range(b1, b1 + b2):
        o = opgrm.pgrm_forward(sd_pgrms[k], ocmm.to_mask(casc.detach()[:, :3]), casc[:, :3], l2[:k - b2], windows=windows)
        l2.append(o)
        casc = o
        tot = tot + ocmm.image_loss(o, hr3, True) * 100
    feat = l1[-1]
    for k in range(b1 - 1, 0, -1):
        ld, feat = ocmm.distill_forward(sd_distill[k - 1], feat, l1[k - 1], True)
        tot = tot + ld * 100
    feat = l2[-1]
    for k in range(b2 - 1, 0, -1):
        ld, feat = ocmm.distill_forward(sd_distill[k + b1 - 2], feat, l2[k - 1], True)
        tot = tot + ld * 100
    o = ocmm.cmm_forward(sd_cmm, l1[-1], l2[-1], True)
    return (tot + ocmm.image_loss(o, hr3, True) * 100) / (b1 + b2 + 1)


def rotate_img(img, arc, rand_offs, off_range=0.2):
    """torch_rotate_img (utils/util.py:37-58) written out at index level: theta = [[cos, sin*r, 0], [-sin/r, cos, 0]],
    r = H/W + (2*rand-1)*off_range (line 44); affine_grid with align_corners=False (base x_j = (2j+1)/W - 1) and
    grid_sample bilinear / zeros padding (lines 55-56, torch defaults).  Pinned by tests/golden/rotate.npz."""
    import torch
    N, C, H, W = img.shape
    ratio = H / float(W) + rand_offs.float() * off_range * 2 - off_range
    c, s = torch.cos(arc.float()), torch.sin(arc.float())
    bx = ((2 * torch.arange(W, dtype=torch.float32) + 1) / W - 1)[None, None, :]
    by = ((2 * torch.arange(H, dtype=torch.float32) + 1) / H - 1)[None, :, None]
    gx = c[:, None, None] * bx + (s * ratio)[:, None, None] * by
    gy = (-s / ratio)[:, None, None] * bx + c[:, None, None] * by
    ix, iy = ((gx + 1) * W - 1) / 2, ((gy + 1) * H - 1) / 2
    x0, y0 = torch.floor(ix), torch.floor(iy)
    out = torch.zeros_like(img, dtype=torch.float32)
    flat = img.float().reshape(N, C, H * W)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            wgt = (1 - (ix - xx).abs()) * (1 - (iy - yy).abs())
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            lin = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().reshape(N, 1, H * W).expand(N, C, H * W)
            out += (flat.gather(2, lin) * (wgt * ok).reshape(N, 1, H * W)).reshape(N, C, H, W)
    return out

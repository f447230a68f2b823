#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the READ-ONLY reference (/root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  Weights and
inputs are NOT stored: both sides regenerate them from dpmn_amd.utils.synth (name-seeded
uniform draws), so a fixture holds only the reference's outputs, the state-dict manifest
(names + shapes) and a float64 checksum of the synthetic weights.

usage: python tools/gen_golden.py [pgrm|parts|cmm|distill|loss|metrics|tsrn|tatt|tbsrn|step|all]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shims  # noqa: E402

ref_shims.install()
from dpmn_amd.utils import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)
torch.set_num_threads(8)


def manifest(sd):
    return np.array(["%s|%s|%s" % (k, ",".join(map(str, v.shape)), str(v.dtype).replace("torch.", ""))
                     for k, v in sd.items()])


def checksum(sd):
    skip = ("relative_position_index", "attn_mask", "num_batches_tracked")
    return float(sum(v.double().sum().item() for k, v in sd.items()
                     if torch.is_floating_point(v) and not any(s in k for s in skip) and not k.endswith(".pe")))


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in arrs.items()})
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def pgrm_args(n=6, dim=96, windows=(2, 4, 8), heads=6):
    return dict(patch_size=[2] * n, embed_dim=[dim] * n, depths=[1] * n, num_heads=[[heads]] * n,
                window_size=[list(windows)] * n, mlp_ratio=[4.] * n, drop_rate=[0.] * n,
                attn_drop_rate=[0.] * n, drop_path_rate=[0.] * n)


# ------------------------------------------------------------------------------------ PGRM
def gen_pgrm():
    from model import pgrm
    B = 2
    for tag, it, mode in (("mode0_iter0", 0, False), ("mode1_iter2", 2, True)):
        m = pgrm.PGRM(iter=it, mode=mode, hidden_size=3, **pgrm_args()).eval()
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=11 + it)
        m.load_state_dict(sd)
        cq = 3 if mode else 2
        if mode:
            x_q = (synth.uniform("x_q", (B, 1, 32, 128), 0, 1, 5) > 0.5).float().repeat(1, 3, 1, 1)
        else:
            x_q = torch.floor(synth.uniform("x_q", (B, cq, 32, 128), 0, 256, 5))
        x_kv = synth.uniform("x_kv", (B, 3, 32, 128), 0, 1, 5)
        res = [synth.uniform("res%d" % i, (B, 3, 32, 128), 0, 1, 5) for i in range(it)]
        out = m(x_q, x_kv, res)
        save("pgrm_" + tag, out=out, manifest=manifest(sd), checksum=checksum(sd),
             meta=np.array([B, it, int(mode), 11 + it, 5]))


def gen_parts():
    """Sub-module pins: WindowAttention (cat tensor fed to SKConv, shift 0 and shifted), SKConv,
    Mlp at config-1 dims, and one BasicLayer at the stress dims."""
    from model import pgrm
    import torch.nn as nn
    B, H, W, C = 1, 16, 64, 96
    for tag, shift in (("shift0", [0, 0, 0]), ("shifted", [1, 2, 4])):
        wa = pgrm.WindowAttention(C, window_size=[2, 4, 8], shift_size=list(shift), num_heads=6,
                                  act_layer=nn.GELU, input_resolution=(H, W)).eval()
        sd = wa.state_dict()
        synth.synth_fill_(sd, seed=21)
        wa.load_state_dict(sd)
        grabbed = {}
        wa.sknet.register_forward_hook(lambda mod, inp, out: grabbed.__setitem__("cat", inp[0].clone()))
        xq = synth.uniform("wa_xq", (B, H, W, C), -1, 1, 6)
        xkv = synth.uniform("wa_xkv", (B, H, W, C), -1, 1, 6)
        out = wa(xq, xkv)
        save("wattn_" + tag, cat=grabbed["cat"].reshape(B, H * W, C), out=out.contiguous(),
             manifest=manifest(sd), checksum=checksum(sd))
    mlp = pgrm.Mlp(C, 4 * C).eval()
    sd = mlp.state_dict()
    synth.synth_fill_(sd, seed=22)
    mlp.load_state_dict(sd)
    x = synth.uniform("mlp_x", (2, H * W, C), -1, 1, 6)
    save("mlp", out=mlp(x), manifest=manifest(sd), checksum=checksum(sd))
    # stress dims: dim 192, windows 4/8/16, 32x128 tokens, B=1 (quirk Q7: only pinned at layer level)
    Hs, Ws, Cs = 32, 128, 192
    bl = pgrm.BasicLayer(Cs, (Hs, Ws), depth=2, num_heads=6, window_size=[4, 8, 16], mlp_ratio=4.,
                         drop=0., attn_drop=0., drop_path=0.).eval()
    sd = bl.state_dict()
    synth.synth_fill_(sd, seed=23)
    bl.load_state_dict(sd)
    xq = synth.uniform("bl_xq", (1, Hs * Ws, Cs), -1, 1, 6)
    xkv = synth.uniform("bl_xkv", (1, Hs * Ws, Cs), -1, 1, 6)
    _, o = bl(xq, xkv)
    save("basiclayer_stress", out=o[:, ::7].contiguous(), manifest=manifest(sd), checksum=checksum(sd))


# ------------------------------------------------------------------------------------ CMM etc.
def gen_cmm():
    from model import cmm
    B = 2
    x1 = synth.uniform("cmm_x1", (B, 3, 32, 128), 0, 1, 7)
    x2 = synth.uniform("cmm_x2", (B, 3, 32, 128), 0, 1, 7)
    for cnum in (8, 64):
        m = cmm.ComplementationModulationModule(cnum=cnum)
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=31)
        sd = {k: v.clone() for k, v in sd.items()}  # state_dict() aliases live buffers (BN stats mutate)
        m.load_state_dict(sd)
        outs = {}
        for training in (False, True):
            m.train(training)
            outs["out_train" if training else "out_eval"] = m(x1, x2)
        save("cmm_cnum%d" % cnum, manifest=manifest(sd), checksum=checksum(sd), **outs)


def gen_distill():
    from model import distill_module
    B = 2
    xd = synth.uniform("dist_deep", (B, 3, 32, 128), 0, 1, 8)
    xs = synth.uniform("dist_shallow", (B, 3, 32, 128), 0, 1, 8)
    m = distill_module.DistillModule()
    sd = m.state_dict()
    synth.synth_fill_(sd, seed=32)
    sd = {k: v.clone() for k, v in sd.items()}
    m.load_state_dict(sd)
    m.train()
    loss, feat = m(xd, xs)
    m.load_state_dict(sd)
    m.eval()
    loss_e, feat_e = m(xd, xs)
    save("distill", loss_train=loss, feat_train=feat, loss_eval=loss_e, feat_eval=feat_e,
         manifest=manifest(sd), checksum=checksum(sd))


def gen_loss():
    from loss import image_loss
    sys.modules.setdefault("IPython", sys.modules["IPython"])
    from utils import ssim_psnr
    B = 4
    a = synth.uniform("loss_a", (B, 3, 32, 128), 0, 1, 9)
    b = synth.uniform("loss_b", (B, 4, 32, 128), 0, 1, 9)
    b3 = (0.7 * a + 0.3 * b[:, :3])
    crit = image_loss.ImageLoss(gradient=True, loss_weight=[1, 1])
    crit_ng = image_loss.ImageLoss(gradient=False, loss_weight=[1, 1])
    save("loss_metrics", loss_grad=crit(a, b3), loss_nograd=crit_ng(a, b3),
         psnr=ssim_psnr.calculate_psnr(a, torch.cat([b3, b[:, 3:]], 1)),
         ssim=ssim_psnr.SSIM()(a, torch.cat([b3, b[:, 3:]], 1)),
         gradmap=image_loss.GradientPriorLoss.gradient_map(a)[:1])


def gen_rotate():
    """torch_rotate_img (utils/util.py:37-58) through the reference function itself (rotate_train = 5 degrees)."""
    from utils import util
    N = 4
    img = synth.uniform("rot_img", (N, 4, 16, 64), 0, 1, 11)
    deg = synth.uniform("rot_deg", (N,), -5.0, 5.0, 11)
    arc = deg / 180.0 * float(np.pi)
    offs = synth.uniform("rot_off", (N,), 0, 1, 11)
    save("rotate", out_lr=util.torch_rotate_img(img, arc, offs),
         out_hr=util.torch_rotate_img(synth.uniform("rot_img_hr", (N, 4, 32, 128), 0, 1, 11), arc, offs))


def gen_tbsrn(x):
    """TBSRN in eval mode (config 3's PSN).  FeatureEnhancer hard-codes .cuda() (tbsrn.py:83): identity on this CPU box."""
    from model import tbsrn
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        m = tbsrn.TBSRN(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32).eval()
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=43)
        sd = {k: v.clone() for k, v in sd.items()}
        m.load_state_dict(sd)
        feats = {}
        m.block2.feature_enhancer.register_forward_hook(lambda mod, i, o: feats.setdefault("fe", o))
        out = m(x)
        save("tbsrn", out=out, fe_block2=feats["fe"][:, ::8].contiguous(), manifest=manifest(sd), checksum=checksum(sd))
    finally:
        torch.Tensor.cuda = orig


def gen_stn_layout():
    """state_dict layout + constructor values of the --STN front end (the README's train/test commands pass --STN)."""
    from model import tatt
    m = tatt.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=True, mask=True, srb_nums=5, hidden_units=32)
    sd = m.state_dict()
    save("stn_layout", manifest=manifest(sd), inverse_kernel=sd["tps.inverse_kernel"], target_control_points=sd["tps.target_control_points"],
         target_coordinate_repr=sd["tps.target_coordinate_repr"][::37].contiguous(), fc2_bias=sd["stn_head.stn_fc2.bias"],
         fc2_weight_absmax=sd["stn_head.stn_fc2.weight"].abs().max())


def gen_stn_fwd():
    """Row a15 forward: the imported STNHead (train and eval mode) and TPSSpatialTransformer on synthetic weights."""
    from model import stn_head, tps_spatial_transformer as tps
    B = 6
    x = synth.uniform("stn_x", (B, 4, 16, 64), 0, 1, 51)
    out = {}
    for mode in ("train", "eval"):
        m = stn_head.STNHead(in_planes=4, num_ctrlpoints=20, activation='none')
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=52)
        sd = {k: v.clone() for k, v in sd.items()}
        m.load_state_dict(sd)
        m.train(mode == "train")
        feat, ctrl = m(x)
        out[mode + "_feat"], out[mode + "_ctrl"] = feat, ctrl
        if mode == "train":
            sd2 = m.state_dict()
            for k in sd2:
                if "running" in k or "num_batches" in k:
                    out["after." + k] = sd2[k].clone()
    t = tps.TPSSpatialTransformer(output_image_size=(16, 64), num_control_points=20, margins=(0.05, 0.05))
    # control points around the identity layout, pushed far enough that some source coordinates leave [0, 1] (clamp path)
    ctrl = t.target_control_points[None] + synth.uniform("stn_ctrl", (B, 20, 2), -0.12, 0.12, 53)
    warped, src = t(x, ctrl)
    save("stn_fwd", manifest=manifest(sd), checksum=checksum(sd), tps_out=warped, tps_src=src[:, ::7].contiguous(),
         tps_src_minmax=np.array([float(src.min()), float(src.max())]), **out)


def gen_psn():
    """PSN backbones in eval mode (frozen in DPMN, super_resolution.py:56-59): TSRN and TATT."""
    from model import tsrn, tatt
    b = synth.synth_batch(2, seed=2)
    x, lv = b["images_lr"], b["label_vecs"]
    m = tsrn.TSRN(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32).eval()
    sd = m.state_dict()
    synth.synth_fill_(sd, seed=41)
    sd = {k: v.clone() for k, v in sd.items()}
    m.load_state_dict(sd)
    save("tsrn", out=m(x), manifest=manifest(sd), checksum=checksum(sd))
    m = tatt.TSRN_TL_TRANS(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32).eval()
    sd = m.state_dict()
    synth.synth_fill_(sd, seed=42)
    sd = {k: v.clone() for k, v in sd.items()}
    m.load_state_dict(sd)
    out, prw = m(x, lv)
    gen_tbsrn(x)
    save("tatt", out=out, pr_weights=prw[:, ::16].contiguous(), tp_map=m.block["1"][:1, :8].contiguous(),
         manifest=manifest(sd), checksum=checksum(sd))


def gen_tpgsr():
    """--arch tpgsr: TSRN_TL (tsrn.py:153-247) in eval mode on the same inputs as the TATT fixture + the InfoGen map."""
    from model import tsrn
    b = synth.synth_batch(2, seed=2)
    x, lv = b["images_lr"], b["label_vecs"]
    m = tsrn.TSRN_TL(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32).eval()
    sd = m.state_dict()
    synth.synth_fill_(sd, seed=44)
    sd = {k: v.clone() for k, v in sd.items()}
    m.load_state_dict(sd)
    save("tpgsr", out=m(x, lv), info=m.infoGen(lv)[:, :, 0, ::7].contiguous(), manifest=manifest(sd), checksum=checksum(sd))


def gen_stack():
    """config 0 (BASELINE.json): TSRN PSN + 1+1 PGRM + CMM, B=4, eval forward of
    interfaces/super_resolution.py:370-449 re-stated around the IMPORTED reference modules.  toMask needs
    torchvision (absent) -> the PIL-pinned restatement oracle.cmm.to_mask is used for the mask prior."""
    from model import tsrn, pgrm, cmm
    from oracle import cmm as ocmm
    B, b1, b2, alpha = 4, 1, 1, 0.5
    batch = synth.synth_batch(B, seed=2)
    psn = tsrn.TSRN(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32).eval()
    mods = [pgrm.PGRM(iter=0, mode=False, hidden_size=3, **pgrm_args(2)).eval(),
            pgrm.PGRM(iter=1, mode=True, hidden_size=3, **pgrm_args(2)).eval()]
    fuse = cmm.ComplementationModulationModule().eval()
    for i, m in enumerate([psn] + mods + [fuse]):
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=100 + i)
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
    prior = torch.floor(synth.uniform("text_prior_0", (B, 2, 32, 128), 0.0, 256.0, 2))
    lr_psn = psn(batch["images_lr"])
    cascade, l1 = lr_psn, []
    for k in range(b1):
        sr = mods[k](prior, cascade[:, :3, :], l1[:k])
        l1.append(sr)
        cascade = sr
    cascade, l2 = lr_psn, []
    for k in range(b1, b1 + b2):
        sr = mods[k](ocmm.to_mask(cascade[:, :3]), cascade[:, :3, :], l2[:(k - b2)])
        l2.append(sr)
        cascade = sr
    out = alpha * fuse(l1[-1], l2[-1]) + (1 - alpha) * lr_psn[:, :3, :, :]
    sys.modules.setdefault("IPython", sys.modules["IPython"])
    from utils import ssim_psnr
    save("stack_cfg0", out=out, psn=lr_psn, branch1=l1[-1], branch2=l2[-1],
         psnr=ssim_psnr.calculate_psnr(out, batch["images_hr"]), ssim=ssim_psnr.SSIM()(out, batch["images_hr"]))


# ------------------------------------------------------------------------------------ gradients of the reference itself
def grad_arrays(prefix, named_grads):
    """npz entries for a set of gradients: helpers.grad_digest (full tensor up to 4096 elements, otherwise norm / sum /
    64 strided samples / one seeded projection) -- the reference's own autograd results, SURVEY.md section 8c."""
    from tests.helpers import grad_digest
    out = {}
    for name, g in named_grads:
        for k, v in grad_digest(name, g).items():
            out["%s%s::%s" % (prefix, name, k)] = v
    return out


def gen_grads():
    """d(out . r)/d(inputs, params) for a fixed seeded cotangent r through the IMPORTED reference modules (train mode,
    drop rates 0): PGRM (both fixture variants), CMM (cnum 8 and 64, batch-statistics BatchNorm), DistillModule."""
    from model import pgrm, cmm, distill_module
    with torch.enable_grad():
        B = 2
        for tag, it, mode in (("mode0_iter0", 0, False), ("mode1_iter2", 2, True)):
            m = pgrm.PGRM(iter=it, mode=mode, hidden_size=3, **pgrm_args()).train()
            sd = m.state_dict()
            synth.synth_fill_(sd, seed=11 + it)
            m.load_state_dict(sd)
            if mode:
                x_q = (synth.uniform("x_q", (B, 1, 32, 128), 0, 1, 5) > 0.5).float().repeat(1, 3, 1, 1)
            else:
                x_q = torch.floor(synth.uniform("x_q", (B, 2, 32, 128), 0, 256, 5))
            x_kv = synth.uniform("x_kv", (B, 3, 32, 128), 0, 1, 5).requires_grad_(True)
            res = [synth.uniform("res%d" % i, (B, 3, 32, 128), 0, 1, 5).requires_grad_(True) for i in range(it)]
            cot = synth.uniform("cot", (B, 3, 32, 128), -1, 1, 5)
            out = m(x_q, x_kv, res)
            (out * cot).sum().backward()
            named = [("x_kv", x_kv.grad)] + [("res%d" % i, r.grad) for i, r in enumerate(res) if r.grad is not None]
            named += [(n, p.grad) for n, p in m.named_parameters() if p.grad is not None]
            save("grads_pgrm_" + tag, out=out.detach(), **grad_arrays("", named))
        x1 = synth.uniform("cmm_x1", (B, 3, 32, 128), 0, 1, 7).requires_grad_(True)
        x2 = synth.uniform("cmm_x2", (B, 3, 32, 128), 0, 1, 7).requires_grad_(True)
        cot = synth.uniform("cmm_cot", (B, 3, 32, 128), -1, 1, 7)
        for cnum in (8, 64):
            m = cmm.ComplementationModulationModule(cnum=cnum).train()
            sd = m.state_dict()
            synth.synth_fill_(sd, seed=31)
            m.load_state_dict({k: v.clone() for k, v in sd.items()})
            x1.grad = x2.grad = None
            out = m(x1, x2)
            (out * cot).sum().backward()
            named = [("x1", x1.grad), ("x2", x2.grad)] + [(n, p.grad) for n, p in m.named_parameters()]
            save("grads_cmm_cnum%d" % cnum, out=out.detach(), **grad_arrays("", named))
        xd = synth.uniform("dist_deep", (B, 3, 32, 128), 0, 1, 8).requires_grad_(True)
        xs = synth.uniform("dist_shallow", (B, 3, 32, 128), 0, 1, 8).requires_grad_(True)
        cot = synth.uniform("dist_cot", (B, 3, 32, 128), -0.01, 0.01, 8)
        m = distill_module.DistillModule().train()
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=32)
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
        loss, feat = m(xd, xs)
        (loss.sum() * 100 + (feat * cot).sum()).backward()
        named = [("xd", xd.grad), ("xs", xs.grad)] + [(n, p.grad) for n, p in m.named_parameters()]
        save("grads_distill", loss=loss.detach(), **grad_arrays("", named))


def gen_step():
    """One optimisation step of interfaces/super_resolution.py:140-278 re-stated around the IMPORTED reference modules
    (TSRN PSN frozen in eval, 2+2 PGRM, 2 DistillModules, CMM, B=2, drop rates 0; text priors are inputs, toMask is the
    PIL-pinned restatement): the loss, every cascade image, and per model the gradient norm that clip_grad_norm_(0.25)
    sees plus digests of the pre-clip gradients.  Same seeds as tests/test_gpu_train.py::test_full_train_step_*."""
    from model import tsrn, pgrm, cmm, distill_module
    from loss import image_loss
    from oracle import cmm as ocmm
    B, b1, b2 = 2, 2, 2
    with torch.enable_grad():
        psn = tsrn.TSRN(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32).eval()
        mods = [pgrm.PGRM(iter=k, mode=False, hidden_size=3, **pgrm_args(4)).train() for k in range(b1)]
        mods += [pgrm.PGRM(iter=k, mode=True, hidden_size=3, **pgrm_args(4)).train() for k in range(b1, b1 + b2)]
        mods.append(cmm.ComplementationModulationModule().train())
        distill = [distill_module.DistillModule().train() for _ in range(b1 + b2 - 2)]
        for i, m in enumerate([psn] + mods + distill):
            sd = m.state_dict()
            synth.synth_fill_(sd, seed=300 + i)
            m.load_state_dict({k: v.clone() for k, v in sd.items()})
        crit = image_loss.ImageLoss(gradient=True, loss_weight=[1, 1])
        batch = synth.synth_batch(B, seed=4)
        priors = [torch.floor(synth.uniform("tp%d" % k, (B, 2, 32, 128), 0, 256, 4)) for k in range(b1)]
        hr = batch["images_hr"]
        with torch.no_grad():
            lr_psn = psn(batch["images_lr"])
        loss = 0
        casc, l1 = lr_psn, []
        for k in range(b1):
            sr = mods[k](priors[k], casc[:, :3, :], l1[:k]); l1.append(sr); casc = sr
            loss = loss + crit(sr, hr[:, :3, :]).mean() * 100
        casc, l2 = lr_psn, []
        for k in range(b1, b1 + b2):
            sr = mods[k](ocmm.to_mask(casc.detach()[:, :3]), casc[:, :3, :], l2[:(k - b2)]); l2.append(sr); casc = sr
            loss = loss + crit(sr, hr[:, :3, :]).mean() * 100
        feat = l1[-1]
        for k in range(b1 - 1, 0, -1):
            ld, feat = distill[k - 1](feat, l1[k - 1]); loss = loss + ld.sum() * 100
        feat = l2[-1]
        for k in range(b2 - 1, 0, -1):
            ld, feat = distill[k + b1 - 2](feat, l2[k - 1]); loss = loss + ld.sum() * 100
        out = mods[-1](l1[-1], l2[-1])
        loss = (loss + crit(out, hr[:, :3, :]).mean() * 100) / (b1 + b2 + 1)
        loss.backward()
        arrs = dict(loss=loss.detach(), psn=lr_psn, cmm_out=out.detach())
        for k in range(b1):
            arrs["branch1_%d" % k] = l1[k].detach()
        for k in range(b2):
            arrs["branch2_%d" % k] = l2[k].detach()
        norms = []
        for i, m in enumerate(mods + distill):
            named = [(n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()]
            norms.append(float(torch.sqrt(sum((g.double() ** 2).sum() for _, g in named))))
            arrs.update(grad_arrays("m%d/" % i, named))
        arrs["grad_norms"] = np.array(norms)
        save("step_tsrn_2p2", **arrs)

# ------------------------------------------------------------------------------------ float64 adjudication of the gradient fixtures
def f64_arrays(prefix, named64, named32):
    """Digests of the float64 gradients plus, per tensor, `ref32_err`: how far the reference's OWN fp32 gradient is from that
    float64 result under the metric the tests use (helpers.grad_error_vs_fixture) -- the yardstick for the HIP path's error."""
    from tests.helpers import grad_digest, grad_error_vs_fixture
    out = {}
    for name, g in named64:
        for k, v in grad_digest(name, g, f64=True).items():
            out["%s%s::%s" % (prefix, name, k)] = v

    class _Npz(dict):
        files = property(lambda self: list(self.keys()))
    z = _Npz(out)
    for name, g32 in named32:
        err, _ = grad_error_vs_fixture(z, prefix + name, g32)
        out["%s%s::ref32_err" % (prefix, name)] = np.array(err)
    return out


def gen_f64():
    """The reference modules in float64 (`.double()`, the SAME fp32 synthetic weights and inputs, exactly representable) as the
    arbiter of the two loose gradient fixtures: grads_cmm_cnum64 (B = 2: 8-sample BatchNorm statistics at the 1 x 4 bottleneck
    make the fp32 gradients ill-conditioned) and step_tsrn_2p2.  A fixture holds float64 digests and, per tensor, the error of
    the reference's own fp32 gradient against them; the GPU tests require err(HIP, f64) <= 1.5 x err(reference fp32, f64)."""
    from model import tsrn, pgrm, cmm, distill_module
    from loss import image_loss
    from oracle import cmm as ocmm
    with torch.enable_grad():
        B = 2
        x1 = synth.uniform("cmm_x1", (B, 3, 32, 128), 0, 1, 7)
        x2 = synth.uniform("cmm_x2", (B, 3, 32, 128), 0, 1, 7)
        cot = synth.uniform("cmm_cot", (B, 3, 32, 128), -1, 1, 7)
        named = {}
        for dt in (torch.float32, torch.float64):
            m = cmm.ComplementationModulationModule(cnum=64).train()
            sd = m.state_dict()
            synth.synth_fill_(sd, seed=31)
            m.load_state_dict({k: v.clone() for k, v in sd.items()})
            m = m.to(dt)
            a, b = x1.detach().clone().to(dt).requires_grad_(True), x2.detach().clone().to(dt).requires_grad_(True)
            out = m(a, b)
            (out * cot.to(dt)).sum().backward()
            named[dt] = [("x1", a.grad), ("x2", b.grad)] + [(n, p.grad) for n, p in m.named_parameters()]
            if dt == torch.float64:
                out64 = out.detach()
        # kink-aware adjudication (tests/helpers.py "Kink-aware float64 adjudication"): the pre-activations within fp32 round-off of a
        # LeakyReLU / ReLU kink, the sign the REFERENCE's fp32 run has there, and the reference's fp32 gradient error against the
        # float64 gradients that differentiate the same branch at those elements (`ref32_err_adj`)
        from tests import helpers as th
        sd32 = {k: v.clone() for k, v in sd.items()}
        sd64 = {k: (v.double() if torch.is_floating_point(v) else v) for k, v in sd32.items()}
        sites64 = th.cmm_sites(sd64, x1.double(), x2.double())
        kinks = th.ambiguous_kinks(sites64)
        # the reference's fp32 pre-activations: inputs of its activation modules (decoder inputs are channel concatenations of the
        # oracle's parts, in the oracle's order)
        m32 = cmm.ComplementationModulationModule(cnum=64).train()
        m32.load_state_dict({k: v.clone() for k, v in sd32.items()})
        ins = {}
        for n_, mod in m32.named_modules():
            if isinstance(mod, (torch.nn.LeakyReLU, torch.nn.ReLU)):
                mod.register_forward_pre_hook(lambda _m, a, n_=n_: ins.__setitem__(n_, a[0].detach()))
        with torch.no_grad():
            m32(x1, x2)
        parts32, off = {}, {}
        for site, x in sites64:
            cons = site.split(">")[1]
            o = off.get(cons, 0)
            parts32[site] = ins[cons][:, o:o + x.shape[1]]
            off[cons] = o + x.shape[1]
            assert parts32[site].shape == x.shape, (site, parts32[site].shape, x.shape)
        forced = [(site, i, float(parts32[site].reshape(-1)[i]) > 0) for site, i, _ in kinks]
        g64_adj = th.cmm_grads_f64(sd32, x1, x2, cot, forced)
        g64_plain = th.cmm_grads_f64(sd32, x1, x2, cot)
        extra = {"kink_site": np.array([k[0] for k in kinks]), "kink_index": np.array([k[1] for k in kinks], dtype=np.int64),
                 "kink_y64": np.array([k[2] for k in kinks]), "kink_ref32_positive": np.array([f[2] for f in forced])}
        for name, g32 in named[torch.float32]:
            extra[name + "::ref32_err_adj"] = np.array(th.rel_l2(g32, g64_adj[name]))
            # the oracle's float64 gradients ARE the reference's (same float64 math): recorded, and asserted by the tests
            extra[name + "::oracle64_vs_ref64"] = np.array(th.rel_l2(g64_plain[name], dict(named[torch.float64])[name]))
        print("cnum64 f64: %d ambiguous pre-activations (|y| < %g); reference fp32 takes the other branch at %d of them; oracle f64 vs reference f64 max %.1e" % (
            len(kinks), th.KINK_TOL, sum(1 for (s_, i, y), f in zip(kinks, forced) if (y > 0) != f[2]), max(float(v) for k, v in extra.items() if k.endswith("oracle64_vs_ref64"))))
        save("grads_cmm_cnum64_f64", out=out64.numpy(), **f64_arrays("", named[torch.float64], named[torch.float32]), **extra)

        b1 = b2 = 2
        batch = synth.synth_batch(B, seed=4)
        priors = [torch.floor(synth.uniform("tp%d" % k, (B, 2, 32, 128), 0, 256, 4)) for k in range(b1)]
        res, masks = {}, None
        for dt in (torch.float32, torch.float64):
            psn = tsrn.TSRN(scale_factor=2, width=128, height=32, STN=False, mask=True, srb_nums=5, hidden_units=32).eval()
            mods = [pgrm.PGRM(iter=k, mode=False, hidden_size=3, **pgrm_args(4)).train() for k in range(b1)]
            mods += [pgrm.PGRM(iter=k, mode=True, hidden_size=3, **pgrm_args(4)).train() for k in range(b1, b1 + b2)]
            mods.append(cmm.ComplementationModulationModule().train())
            distill = [distill_module.DistillModule().train() for _ in range(b1 + b2 - 2)]
            for i, m in enumerate([psn] + mods + distill):
                sd = m.state_dict()
                synth.synth_fill_(sd, seed=300 + i)
                m.load_state_dict({k: v.clone() for k, v in sd.items()})
                m.to(dt)
            crit = image_loss.ImageLoss(gradient=True, loss_weight=[1, 1])
            hr = batch["images_hr"].to(dt)
            with torch.no_grad():
                lr_psn = psn(batch["images_lr"].to(dt))
            loss = 0
            casc, l1 = lr_psn, []
            for k in range(b1):
                sr = mods[k](priors[k].to(dt), casc[:, :3, :], l1[:k]); l1.append(sr); casc = sr
                loss = loss + crit(sr, hr[:, :3, :]).mean() * 100
            casc, l2 = lr_psn, []
            new_masks = []
            for k in range(b1, b1 + b2):
                # the mask prior is a threshold (discontinuous): the float64 run takes the fp32 run's masks, so that both
                # differentiate the same function
                mk = ocmm.to_mask(casc.detach()[:, :3].float()) if masks is None else masks[k - b1]
                new_masks.append(mk)
                sr = mods[k](mk.to(dt), casc[:, :3, :], l2[:(k - b2)]); l2.append(sr); casc = sr
                loss = loss + crit(sr, hr[:, :3, :]).mean() * 100
            masks = masks or new_masks
            feat = l1[-1]
            for k in range(b1 - 1, 0, -1):
                ld, feat = distill[k - 1](feat, l1[k - 1]); loss = loss + ld.sum() * 100
            feat = l2[-1]
            for k in range(b2 - 1, 0, -1):
                ld, feat = distill[k + b1 - 2](feat, l2[k - 1]); loss = loss + ld.sum() * 100
            out = mods[-1](l1[-1], l2[-1])
            loss = (loss + crit(out, hr[:, :3, :]).mean() * 100) / (b1 + b2 + 1)
            loss.backward()
            res[dt] = (float(loss.detach()), [[(n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()]
                                              for m in mods + distill])
        arrs = dict(loss=np.array(res[torch.float64][0]), loss_ref32_err=np.array(abs(res[torch.float32][0] - res[torch.float64][0]) / abs(res[torch.float64][0])))
        norms, norm_err = [], []
        for i, (n64, n32) in enumerate(zip(res[torch.float64][1], res[torch.float32][1])):
            arrs.update(f64_arrays("m%d/" % i, n64, n32))
            a = float(torch.sqrt(sum((g.double() ** 2).sum() for _, g in n64)))
            b_ = float(torch.sqrt(sum((g.double() ** 2).sum() for _, g in n32)))
            norms.append(a); norm_err.append(abs(a - b_) / a)
        arrs["grad_norms"] = np.array(norms)
        arrs["grad_norms_ref32_err"] = np.array(norm_err)
        save("step_tsrn_2p2_f64", **arrs)


def gen_visionlan():
    """VisionLAN (the branch-1 text-prior recogniser) in eval mode through the imported reference: per-step logits captured at
    Prediction, the flattened (output, out_length) pair MLM_VRM returns, a digest of the backbone feature map and the strings
    cha_encdec.decode (restated: the dict file is read there, utils.py:13-16) produces.  B = 3 images of 64x256."""
    from model.VisionLAN.VisionLAN import VisionLAN
    B = 3
    m = VisionLAN(strides=[(1, 1), (2, 2), (2, 2), (2, 2), (1, 1), (1, 1)], input_shape=[3, 64, 256]).eval()
    pos_tab = m.MLM_VRM.SequenceModeling.position_enc.pos_table[0, ::5, ::7].clone()      # the constructor's sinusoid table
    sd = m.state_dict()
    synth.synth_fill_(sd, seed=61)
    sd = {k: v.clone() for k, v in sd.items()}
    m.load_state_dict(sd)
    x = synth.uniform("vl_img", (B, 3, 64, 256), 0, 1, 62)
    grab = {}
    m.MLM_VRM.Prediction.register_forward_hook(lambda mod, i, o: grab.__setitem__("logits", o.clone()))
    m.backbone.register_forward_hook(lambda mod, i, o: grab.__setitem__("feat", o[-1].clone()))
    m.MLM_VRM.SequenceModeling.register_forward_hook(lambda mod, i, o: grab.__setitem__("enc", o[0].clone()))
    output, out_length = m(x, None, '', False)
    dic = [l.replace("\n", "") for l in open(os.path.join(ref_shims.REF_ROOT, "dic_36.txt")).readlines()]
    prob = torch.softmax(output, 1)
    texts, start = [], 0
    for i in range(B):
        n = int(out_length[i])
        idx = prob[start:start + n].topk(1)[1][:, 0].tolist()
        texts.append("".join(dic[c - 1] if 0 < c <= len(dic) else "" for c in idx))
        start += n
    # the decode loop (VisionLAN.py:107-135) on logits with EOS in the middle / at step 0 / never: Prediction replaced by a stub
    # returning crafted logits, everything after it is the reference's own code
    grab = dict(grab)                                    # (the hooks fire again below)
    m.backbone._forward_hooks.clear(); m.MLM_VRM.SequenceModeling._forward_hooks.clear(); m.MLM_VRM.Prediction._forward_hooks.clear()
    crafted = synth.uniform("vl_crafted", (4, 26, 37), -1, 1, 64)
    crafted[:, :, 0] -= 3.0
    crafted[0, 6, 0] = 5.0; crafted[0, 9, 0] = 5.0      # first EOS at step 6 -> length 7
    crafted[1, 0, 0] = 5.0                               # EOS at step 0 -> length 1, empty string
    crafted[3, 24, 0] = 5.0                              # EOS at the last step -> length 25
    m.MLM_VRM.Prediction.forward = lambda *a, **k: crafted.clone()
    d_out, d_len = m(synth.uniform("vl_img4", (4, 3, 64, 256), 0, 1, 62), None, '', False)
    d_prob = torch.softmax(d_out, 1)
    d_texts, start = [], 0
    for i in range(4):
        n = int(d_len[i])
        idx = d_prob[start:start + n].topk(1)[1][:, 0].tolist()
        d_texts.append("".join(dic[c - 1] if 0 < c <= len(dic) else "" for c in idx))
        start += n
    save("visionlan", pos_table_sample=pos_tab, decode_output=d_out, decode_length=d_len, decode_texts=np.array(d_texts), logits=grab["logits"], output=output, out_length=out_length, feat_sample=grab["feat"][:, ::16, :, ::4].contiguous(),
         enc_sample=grab["enc"][:, ::8, ::8].contiguous(), texts=np.array(texts), dict36=np.array("".join(dic)),
         manifest=manifest(sd), checksum=checksum(sd))


GENS = {"f64": gen_f64, "tpgsr": gen_tpgsr, "visionlan": gen_visionlan, "grads": gen_grads, "step": gen_step, "stack": gen_stack, "psn": gen_psn, "pgrm": gen_pgrm, "parts": gen_parts, "cmm": gen_cmm, "distill": gen_distill, "loss": gen_loss, "rotate": gen_rotate, "stn": gen_stn_layout, "stn_fwd": gen_stn_fwd}



def gen_crnn():
    """The frozen CRNN that produces TATT's label_vecs on real data (model/crnn/crnn.py through the import; CRNN_init's
    CRNN(32, 1, 37, 256), base.py:411-417) + the reference lines around it: parse_crnn_data (base.py:419-425) and the softmax /
    permute of super_resolution.py:165-169, on name-seeded synthetic weights and LR images."""
    import torch.nn.functional as F
    from model.crnn import crnn
    m = crnn.CRNN(32, 1, 37, 256).eval()
    sd = m.state_dict()
    synth.synth_fill_(sd, seed=71)
    sd = {k: v.clone() for k, v in sd.items()}
    m.load_state_dict(sd)
    imgs = synth.uniform("crnn_lr", (3, 3, 16, 64), 0, 1, 72)
    x = F.interpolate(imgs, (32, 100), mode='bicubic')                                   # parse_crnn_data
    gray = 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]
    logits = m(gray)
    lv = torch.nn.functional.softmax(logits, -1).permute(1, 0, 2).unsqueeze(1).permute(0, 3, 1, 2)
    save("crnn", logits=logits, label_vecs=lv.contiguous(), manifest=manifest(sd), checksum=checksum(sd))


def gen_collate():
    """Data path: the reference's own `resizeNormalize` and `alignCollate_realWTLAMask.__call__`
    (dataset/dataset.py:1266-1319, 1966-2076) and `str_filt` (utils/util.py) run on five synthetic RGB images of ragged sizes
    (the generator of tests/test_dataset.py::_fake_env, RandomState(7)) and labels that exercise every branch of the label
    spreading (1 char, empty, 2..25 chars, >= 26 chars, out-of-alphabet characters).
    Import-time shims beyond tools/ref_shims.py (none of them executed except ToTensor): lmdb, imgaug.augmenters (constructed by
    alignCollate_syn.__init__, never applied by this class), cv2, torchvision.utils.  torchvision.transforms.ToTensor IS executed
    and is stood in for by its definition for uint8 PIL images (HWC uint8 -> CHW float32 / 255): torchvision is absent."""
    import types
    from PIL import Image

    class _Any:
        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)

    tvt = sys.modules["torchvision.transforms"]
    tvt.ToTensor = ToTensor
    tvt.ToPILImage = _Any
    sys.modules["torchvision"].utils = types.ModuleType("torchvision.utils")
    sys.modules["torchvision.utils"] = sys.modules["torchvision"].utils
    sys.modules["torchvision.utils"].make_grid = None
    sys.modules["torchvision"].transforms = tvt
    for name in ("lmdb", "imgaug"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["imgaug.augmenters"] = _Any()
    sys.modules["imgaug"].augmenters = sys.modules["imgaug.augmenters"]
    import scipy
    scipy.finfo = np.finfo      # version skew: dataset.py builds an (unused here) blur kernel at import through scipy.finfo (removed in scipy 1.x)
    from dataset import dataset as rds       # /root/reference/dataset/dataset.py
    from utils import str_filt as ref_str_filt

    rng = np.random.RandomState(7)
    words = ["Hello", "a", "", "SuperResolution-2023!", "abcdefghijklmnopqrstuvwxyz0123"]
    batch = []
    for i in range(1, 6):
        hr = rng.randint(0, 256, (40 + 3 * i, 150 + 7 * i, 3)).astype(np.uint8)
        lr = rng.randint(0, 256, (18 + i, 70 + 3 * i, 3)).astype(np.uint8)
        ph, pl = Image.fromarray(hr), Image.fromarray(lr)
        batch.append((ph, pl, ph, pl, ref_str_filt(words[i - 1], "all")))
    out = {}
    for mask in (True, False):
        col = rds.alignCollate_realWTLAMask(imgH=32, imgW=128, down_sample_scale=2, mask=mask)
        r = col(batch)
        tag = "mask" if mask else "nomask"
        out["hr_" + tag] = r[0].numpy()
        out["lr_" + tag] = r[2].numpy()
        if mask:
            out["label_vecs"] = r[6].numpy()
            out["weighted_masks"] = r[7].numpy()
            out["weighted_tics"] = r[8].numpy()
            out["label_strs"] = np.array(list(r[5]))
    filt_in = ["Hello, World!", "abc DEF 123", "", "!!", "MiXed-Case_09"]
    for voc in ("lower", "upper", "all", "digit"):
        out["str_filt_" + voc] = np.array([ref_str_filt(w, voc) for w in filt_in])
    out["str_filt_in"] = np.array(filt_in)
    save("collate", **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["all"]
    GENS["crnn"] = gen_crnn
    GENS["collate"] = gen_collate      # last: it installs extra import shims (lmdb, imgaug, torchvision.utils)
    for name, fn in GENS.items():
        if "all" in which or name in which:
            fn()

"""Named workloads of BASELINE.json (synthetic weights + inputs) for bench.py, the smoke
test and the parity tests.  Host code only: builds the reference-compatible module stack and its inputs."""
from types import SimpleNamespace

import torch

from .utils import synth

CONFIGS_TRAIN = {"cfg2": "cfg1"}   # config 2 = the training step on config 1's stack

CONFIGS = {
    # name: (arch, b1, b2, per-GPU batch)
    "cfg0": ("tsrn", 1, 1, 4),    # TSRN + 1+1 PGRM, B=4 (the reference's CPU-runnable plumbing case)
    "cfg1": ("tatt", 3, 3, 48),   # TATT + 3+3 PGRM, embed 96, windows 2/4/8, B=48 fp32 forward (headline metric)
    "cfg3": ("tbsrn", 3, 3, 64),  # TBSRN PSN + 3+3 PGRM, B=64 (config 3 without the out-of-scope in-loop VisionLAN recogniser)
    "cfg4": ("tsrn", 6, 6, 96),   # stress: embed 192, 6+6 PGRM, windows 4/8/16, 32x128 -> 64x256, B=96 (BASELINE.json configs[4])
}
# per-config PGRM geometry: (embed_dim, windows, SR height, SR width).  cfg4 cannot run in the unmodified reference (quirk Q7:
# weight_list is hard-wired to 32x128); here the PGRM's output size follows config.TRAIN.height/width (interfaces/base.py)
GEOM = {"cfg4": (192, (4, 8, 16), 64, 256)}
DEFAULT_GEOM = (96, (2, 4, 8), 32, 128)


def geom(name):
    return GEOM.get(name, DEFAULT_GEOM)


def describe(name):
    """dict(arch, b1, b2, shape, text, windows) for bench.py's JSON line."""
    arch, b1, b2, _ = CONFIGS[name]
    dim, win, h, w = geom(name)
    return dict(arch=arch, b1=b1, b2=b2, shape="%dx%d->%dx%d" % (h // 2, w // 2, h, w), windows=win,
                text="%s PSN + %d+%d PGRM (embed %d, windows %s) + CMM" % (arch.upper(), b1, b2, dim, "/".join(map(str, win))))


def cpu_priors(name, n_img):
    """the same synthetic text priors build() uploads, on the CPU (bench.py's cpu_baseline child)."""
    _, b1, _, _ = CONFIGS[name]
    _, _, h, w = geom(name)
    return [torch.floor(synth.uniform("text_prior_%d" % k, (n_img, 2, h, w), 0.0, 256.0, 2)) for k in range(b1)]


def make_args(arch, b1, b2, batch, drop=0, dim=96, windows=(2, 4, 8)):
    """drop: one rate for --drop_rate / --attn_drop_rate / --drop_path_rate (the reference README's training command uses
    0.1 for all three; 0 = the deterministic configuration the parity tests and the headline bench run)."""
    n = b1 + b2
    rep = lambda v: ",".join([str(v)] * n) + ","
    return SimpleNamespace(
        arch=arch, test=False, test_data_dir=None, batch_size=batch, resume=None, vis_dir=None, rec="aster", mask=True,
        gradient=True, hd_u=32, srb=5, STN=False, patch_size=rep(2), embed_dim=rep(dim), window_size=rep(",".join(map(str, windows))),
        depths=rep(1), num_heads=rep(6), mlp_ratio=rep(4), drop_rate=rep(drop), attn_drop_rate=rep(drop), drop_path_rate=rep(drop),
        rotate_train=0.0, rotate_test=0.0, stu_iter_b1=b1, stu_iter_b2=b2, tpg="visionlan", rec_path=None, font_path=None, synthetic_prior=True,
        sr_share=False, alpha=0.5, window_num=3)


def make_config(batch, height=32, width=128):
    train = SimpleNamespace(batch_size=batch, width=width, height=height, epochs=1, cuda=True, ngpu=1, workers=0, resume="",
                            ckpt_dir="./ckpt", voc_type="all", saveInterval=20, displayInterval=20, lr=0.001,
                            optimizer="Adam", beta1=0.5, manualSeed=2, max_len=100, keep_ratio=False, down_sample_scale=2)
    return SimpleNamespace(TRAIN=train)


def build(name, batch=None, seed=100, device=None, drop=0):
    """Returns (sr: TextSR, models, psn, inputs dict) with name-seeded synthetic weights (module i -> seed+i in the
    order [PSN, PGRM_0.., CMM], identical to tools/gen_golden.py::gen_stack for cfg0)."""
    from .interfaces.super_resolution import TextSR
    arch, b1, b2, B = CONFIGS[name]
    B = batch or B
    dim, win, h, w = geom(name)
    sr = TextSR(make_config(B, h, w), make_args(arch, b1, b2, B, drop, dim, win))
    models, psn = sr.build_models()
    for i, m in enumerate([psn] + models):
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=seed + i)
        m.load_state_dict(sd)
        m.eval()
        for p in m.parameters():
            p.requires_grad = False
    dev = device or sr.device
    batch_d = synth.synth_batch(B, seed=2, h_lr=h // 2, w_lr=w // 2)
    inputs = {k: v.to(dev) for k, v in batch_d.items()}
    inputs["text_priors"] = [torch.floor(synth.uniform("text_prior_%d" % k, (B, 2, h, w), 0.0, 256.0, 2)).to(dev)
                             for k in range(b1)]
    return sr, models, psn, inputs


def build_text_prior(sr, b1, seed=400):
    """The in-loop recogniser-driven text prior (config 3: "VisionLAN text-prior branch enabled") with synthetic recogniser
    weights: b1 VisionLAN mirrors + the glyph atlas -> callable(cascade, k)."""
    from .interfaces.text_prior import VisionLANTextPrior, build_recognizers
    recs = build_recognizers(b1, sr.device, allow_random=True)       # filled with synthetic weights right below
    for i, r in enumerate(recs):
        sd = r.state_dict()
        synth.synth_fill_(sd, seed=seed + i)
        r.load_state_dict(sd)
    return VisionLANTextPrior(recs, sr.device)


def state_dicts_cpu(models, psn):
    return ([{k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in models],
            {k: v.detach().cpu().clone() for k, v in psn.state_dict().items()})


def cpu_state_dicts(workload_name, seed=100):
    """CPU-only synthetic reference-keyed state dicts (same seeding as build()); used by bench.py's cpu_baseline child."""
    from .model.pgrm import PGRM
    from .model.cmm import ComplementationModulationModule
    from .model.tsrn import TSRN
    from .model.tatt import TSRN_TL_TRANS
    from .model.tbsrn import TBSRN
    from .model.tsrn import TSRN_TL
    arch, b1, b2, _ = CONFIGS[workload_name]
    n = b1 + b2
    dim, win, h, w = geom(workload_name)
    args = dict(img_size=[h, w], patch_size=[2] * n, embed_dim=[dim] * n, depths=[1] * n, num_heads=[[6]] * n, window_size=[list(win)] * n,
                mlp_ratio=[4.] * n, drop_rate=[0.] * n, attn_drop_rate=[0.] * n, drop_path_rate=[0.] * n)
    kw = dict(scale_factor=2, width=w, height=h, STN=False, mask=True, srb_nums=5, hidden_units=32)
    psn = {"tatt": TSRN_TL_TRANS, "tbsrn": TBSRN, "tpgsr": TSRN_TL}.get(arch, TSRN)(**kw)
    mods = [PGRM(iter=k, mode=False, hidden_size=3, **args) for k in range(b1)]
    mods += [PGRM(iter=k, mode=True, hidden_size=3, **args) for k in range(b1, b1 + b2)]
    mods.append(ComplementationModulationModule())
    sds = []
    for i, m in enumerate([psn] + mods):
        sd = m.state_dict()
        synth.synth_fill_(sd, seed=seed + i)
        sds.append({k: v.clone() for k, v in sd.items()})
    return arch, b1, b2, sds[0], sds[1:]

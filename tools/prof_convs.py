"""Per-call timing of every ops.conv2d launch in one cfg1 forward (eval): shape, time, TFLOP/s, weight GB/s.
usage: python tools/prof_convs.py [fwd]"""
import os
import sys

os.environ.setdefault("DPMN_CMM_NATIVE", "0")      # the composed per-op CMM path: ops.conv2d is what this tool wraps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops, workload

sr, models, psn, inp = workload.build("cfg1")
step = lambda: sr.refine(models, psn, inp["images_lr"], inp.get("label_vecs"), text_priors=inp["text_priors"])
for _ in range(3):
    step()
torch.cuda.synchronize()
rec = []
orig = ops.conv2d


def timed(inputs, wp, bias, cout, k, *a, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = orig(inputs, wp, bias, cout, k, *a, **kw)
    e.record()
    rec.append((s, e, [tuple(t.shape) for t in inputs], cout, k, kw.get("stride", 1), kw.get("phase"), tuple(out.shape), wp.numel()))
    return out


ops.conv2d = timed
recT = []
origT = ops.convT_s2k4


def timedT(inputs, packs, cout, *a, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = origT(inputs, packs, cout, *a, **kw)
    e.record()
    recT.append((s, e, [tuple(t.shape) for t in inputs], cout))
    return out


ops.convT_s2k4 = timedT
import dpmn_amd.model.cmm as cmm_mod, dpmn_amd.model.pgrm as pgrm_mod, dpmn_amd.model.tsrn as tsrn_mod
step()
torch.cuda.synchronize()
tot = 0.0
for s, e, shp, cout, k, stride, phase, oshp, wn in rec:
    ms = s.elapsed_time(e)
    tot += ms
    cin = sum(x[3] for x in shp)
    kh, kw_ = (k, k) if isinstance(k, int) else k
    B, H, W = shp[0][:3]
    if phase is not None:
        M = B * H * W
    elif len(oshp) == 4:
        M = oshp[0] * oshp[1] * oshp[2] if oshp[3] == cout or oshp[3] * 4 == cout else oshp[0] * oshp[2] * oshp[3]
        if oshp[3] * 4 == cout:
            M //= 4
    fl = 2.0 * M * cin * kh * kw_ * cout
    print("in %-44s cout %4d k %s s%d %-6s M %6d  %7.1f us %6.1f TF  W %6.1f MB %5.2f TB/s" % (
        "+".join("%dx%dx%d" % (x[1], x[2], x[3]) for x in shp), cout, k, stride, phase or "", M, ms * 1e3, fl / ms / 1e9, wn * 4 / 1e6,
        wn * 4 / ms / 1e9))
print("conv total %.2f ms in %d launches" % (tot, len(rec)))
totT = 0.0
for s, e, shp, cout in recT:      # ConvTranspose2d(4,2,1): 4 phases of a 2x2 conv, one launch
    ms = s.elapsed_time(e)
    totT += ms
    cin = sum(x[3] for x in shp)
    B, H, W = shp[0][:3]
    fl = 2.0 * B * H * W * 4 * (4 * cin) * cout
    print("convT in %-38s cout %4d  M/phase %6d K %5d  %7.1f us %6.1f TF" % ("+".join("%dx%dx%d" % (x[1], x[2], x[3]) for x in shp), cout,
                                                                          B * H * W, 4 * cin, ms * 1e3, fl / ms / 1e9))
print("convT total %.2f ms in %d launches" % (totT, len(recT)))

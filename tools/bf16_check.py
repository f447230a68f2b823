"""fp32 vs bf16-operand forward of a workload: max |difference| of every stage, PSNR / SSIM shift.  python tools/bf16_check.py [cfg1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload, ops, _abi
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
sr, models, psn, inp = workload.build(name, batch=int(sys.argv[2]) if len(sys.argv) > 2 else None)
def run():
    return sr.refine(models, psn, inp["images_lr"], inp.get("label_vecs"), text_priors=inp["text_priors"], return_all=True)
_abi.lib.dpmn_set_compute_dtype(0)
o32, m32 = run()
_abi.lib.dpmn_set_compute_dtype(1)
o16, m16 = run()
_abi.lib.dpmn_set_compute_dtype(0)
d = lambda a, b: float((a - b).abs().max())
print("psn", d(m32["psn"], m16["psn"]))
for k in ("branch1", "branch2"):
    print(k, [round(d(a, b), 5) for a, b in zip(m32[k], m16[k])])
print("cmm", d(m32["cmm"], m16["cmm"]), "output", d(o32, o16), "|out| max", float(o32.abs().max()))
p32, s32 = ops.psnr_ssim(o32, inp["images_hr"]); p16, s16 = ops.psnr_ssim(o16, inp["images_hr"])
print("PSNR %.5f vs %.5f  SSIM %.6f vs %.6f" % (float(p32), float(p16), float(s32), float(s16)))

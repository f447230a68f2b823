"""ORACLE (test infrastructure only) -- CPU restatement of the DPMN eval forward stack
(interfaces/super_resolution.py:370-449): frozen PSN -> branch-1 PGRMs (text prior) -> branch-2 PGRMs
(mask prior from toMask) -> CMM -> alpha blend.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file.  Pinned by tests/golden/stack_cfg0.npz (reference modules driven
by a restatement of those lines in tools/gen_golden.py; toMask itself is pinned against PIL).
"""
from . import cmm as ocmm
from . import pgrm as opgrm
from . import tsrn as otsrn


def refine(sd_psn, sd_pgrms, sd_cmm, arch, b1, b2, images_lr, label_vecs, text_priors, alpha=0.5, return_all=False):
    if arch == "tatt":
        psn, _ = otsrn.tatt_forward(sd_psn, images_lr, label_vecs)
    else:
        psn = otsrn.tsrn_forward(sd_psn, images_lr)
    cascade, br1 = psn, []
    for k in range(b1):
        sr = opgrm.pgrm_forward(sd_pgrms[k], text_priors[k], cascade[:, :3], br1[:k])
        br1.append(sr)
        cascade = sr
    cascade, br2 = psn, []
    for k in range(b1, b1 + b2):
        sr = opgrm.pgrm_forward(sd_pgrms[k], ocmm.to_mask(cascade[:, :3]), cascade[:, :3], br2[:(k - b2)])
        br2.append(sr)
        cascade = sr
    fused = ocmm.cmm_forward(sd_cmm, br1[-1], br2[-1], False)
    out = alpha * fused + (1 - alpha) * psn[:, :3]
    if return_all:
        return out, dict(psn=psn, branch1=br1, branch2=br2, cmm=fused)
    return out

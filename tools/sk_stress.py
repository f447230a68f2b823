"""Stream-K conv launch under repetition: max |diff| vs the fixed-split path and run-to-run equality over N runs of every
tests/test_gpu_streamk.py case (tools/sk_stress.py [N])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_streamk as m
from dpmn_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
for affine in (False, True):
    for case in m.CASES:
        ref, run = m._conv_case(dev, *case, affine)
        ops.STREAM_K = False
        old = run()
        ops.STREAM_K = True
        first = run()
        nbad, worst = 0, 0.0
        for i in range(N):
            g = run()
            if not torch.equal(g, first):
                nbad += 1
                worst = max(worst, float((g - first).abs().max()))
        print("aff %d case %-50s |sk-old| %.3e |sk-ref| %.3e  unequal runs %d/%d (max %.3e)" % (
            affine, str(case), float((first - old).abs().max()), float((first.permute(0, 3, 1, 2).cpu() - ref).abs().max()), nbad, N, worst), flush=True)

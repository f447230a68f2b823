"""Runs only the roofline kernel of bench.py (k_gemm_pw at the cfg1 shapes) so that a PMC pass sees nothing else:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python tools/roofline_kernel.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out -- python tools/roofline_kernel.py
(separate passes: FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops
from dpmn_amd.utils import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda:0")
g = synth.uniform("rf_g", (B, 1024, 384), -1, 1, 5).to(dev)
w = synth.uniform("rf_w", (384, 384), -0.1, 0.1, 5).to(dev)
b = synth.uniform("rf_b", (384,), -0.1, 0.1, 5).to(dev)
for _ in range(20):
    ops.pointwise(g, w, b)
torch.cuda.synchronize()

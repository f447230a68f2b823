"""Do torch's OWN elementwise / reduction kernels (compiled by PyTorch, packed fp32 ops or not) change their results next to a bf16x3
implicit-GEMM conv on another stream?  (v_pk_fma_f32 in k_bigru did.)  The ops the training step still issues
(aten add / add_ / mul / div / fill_ / copy_ / cat / sum / mean) on tensors of the step's sizes, 120 launches each beside the aggressor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import ops, workload, _abi
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
_abi.check(_abi.lib.dpmn_set_compute_dtype(2))
sr, models, psn, inp = workload.build("cfg1", batch=6)
P = psn._trunk_pack()
b = synth.synth_batch(6, seed=61)
t7b = psn._head(b["images_lr"].to(dev), P).clone()
ch = t7b.shape[-1]
g = torch.Generator(device="cpu").manual_seed(3)
x = torch.randn(48 * 64 * 32 * 128 // 16, generator=g).to(dev)        # 786 k elements
y = torch.randn(x.numel(), generator=g).to(dev)
big = torch.randn(48, 3, 32, 128, generator=g).to(dev)
victims = {
    "add": lambda: x + y, "mul": lambda: x * y, "div scalar": lambda: x / 7.0, "mul scalar": lambda: x * 100.0,
    "add_ (accumulate)": lambda: x.clone().add_(y), "fused multiply-add (addcmul)": lambda: torch.addcmul(x, x, y),
    "sum": lambda: x.sum(), "mean": lambda: big.mean(), "cat": lambda: torch.cat([big, big * 2.0], 1), "copy_ strided": lambda: big.permute(0, 2, 3, 1).contiguous(),
    "sqrt": lambda: x.abs().sqrt(), "fill + add": lambda: torch.zeros_like(x).add_(x, alpha=0.25),
}
alone = {k: v().clone() for k, v in victims.items()}
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
tot_bad = 0
for name, fn in victims.items():
    bad = 0
    for rep in range(30):
        with torch.cuda.stream(sB):
            for _ in range(6):
                ops.conv2d([t7b], *P["up"], 4 * ch, 3, pad=1, epi_act="mish", pixel_shuffle=True)       # the aggressor: x3 implicit GEMM + split-K reduce
        with torch.cuda.stream(sA):
            outs = [fn() for _ in range(4)]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, alone[name]) else 1 for o in outs)
    tot_bad += bad
    print("%-30s mismatches %3d / 120" % (name, bad))
print("TOTAL mismatches", tot_bad)

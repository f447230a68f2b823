"""Host-side cost of issuing one CMM training forward / backward (called directly in this thread, no GPU sync inside the profiled
region): cProfile over N calls.  Usage: python tools/host_profile_cmm.py [N]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd.model.cmm import ComplementationModulationModule as CMM
from dpmn_amd.train import cmm_train
from dpmn_amd.train.optim import Trainer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
m = CMM(c_img=3, cnum=64).to(dev).train()
for p in m.parameters():
    p.requires_grad = True
tr = Trainer([m], lr=1e-3, beta1=0.5, max_norm=0.25)
B = 48
x1, x2 = torch.rand(B, 3, 32, 128, device=dev), torch.rand(B, 3, 32, 128, device=dev)
cot = torch.rand(B, 3, 32, 128, device=dev)
for phase in ("warm", "timed"):
    tr.zero_grad()
    pf, pb = cProfile.Profile(), cProfile.Profile()
    tf = tb = 0.0
    for _ in range(N):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pf.enable()
        out, graph = cmm_train.build(m, x1, x2)
        pf.disable()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pb.enable()
        cmm_train.backward(m, graph, cot, (True, True))
        pb.disable()
        t3 = time.perf_counter()
        tf += t1 - t0
        tb += t3 - t2
    torch.cuda.synchronize()
print("CMM training forward: host %.0f us per call; backward: host %.0f us per call (under cProfile)" % (tf / N * 1e6, tb / N * 1e6))
for name, p in (("forward", pf), ("backward", pb)):
    print("==", name)
    pstats.Stats(p).sort_stats("tottime").print_stats(16)

"""Host-side cost of issuing one PGRM training forward (no GPU sync inside the profiled region): cProfile over N calls.
Usage: python tools/host_profile_pgrm.py [N]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd.model.pgrm import PGRM
from dpmn_amd.train import pgrm_train

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
n = 6
args = dict(patch_size=[2] * n, embed_dim=[96] * n, depths=[1] * n, num_heads=[[6]] * n, window_size=[[2, 4, 8]] * n,
            mlp_ratio=[4.] * n, drop_rate=[0.1] * n, attn_drop_rate=[0.1] * n, drop_path_rate=[0.1] * n)
m = PGRM(iter=2, mode=False, hidden_size=3, **args).to(dev).train()
B = 48
x_q = torch.rand(B, 2, 32, 128, device=dev)
x_kv = torch.rand(B, 3, 32, 128, device=dev).requires_grad_(True)
res = [torch.rand(B, 3, 32, 128, device=dev) for _ in range(2)]
for native in (True, False):
    pgrm_train.NATIVE_FWD = native
    for _ in range(5):
        out = m(x_q, x_kv, res)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        out = m(x_q, x_kv, res)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("native=%s: host %.1f us per forward (GPU-inclusive %.1f us)" % (native, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N):
        out = m(x_q, x_kv, res)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)

# ---- the backward's issue cost, called directly in this thread (autograd runs it in its device thread, invisible to cProfile)
from dpmn_amd.train.optim import Trainer
tr = Trainer([m], lr=1e-3, beta1=0.5, max_norm=0.25)
pgrm_train.NATIVE_FWD = True
dout = torch.rand(B, 3, 32, 128, device=dev)
for label in ("warm", "timed"):
    tr.zero_grad()
    svs = [pgrm_train.forward(m, x_q, x_kv.detach(), res, pgrm_train.drop_config(m))[1] for _ in range(N)]
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for sv in svs:
        pgrm_train.backward_deferred(m, sv, dout, True)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    del svs
print("backward: host %.1f us per call (under cProfile)" % ((t1 - t0) / N * 1e6))
pstats.Stats(pr).sort_stats("tottime").print_stats(25)

"""Time one PGRM forward at the config-1 batch (B=48) -- profiling helper for rocprofv3."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd.model.pgrm import PGRM
from dpmn_amd.utils import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
n = 6
args = dict(patch_size=[2] * n, embed_dim=[96] * n, depths=[1] * n, num_heads=[[6]] * n, window_size=[[2, 4, 8]] * n,
            mlp_ratio=[4.] * n, drop_rate=[0.] * n, attn_drop_rate=[0.] * n, drop_path_rate=[0.] * n)
m = PGRM(iter=0, mode=False, hidden_size=3, **args).eval()
sd = m.state_dict(); synth.synth_fill_(sd, 1); m.load_state_dict(sd)
dev = torch.device("cuda:0")
m = m.to(dev)
xq = torch.floor(synth.uniform("xq", (B, 2, 32, 128), 0, 256, 1)).to(dev)
xkv = synth.uniform("xkv", (B, 3, 32, 128), 0, 1, 1).to(dev)
with torch.no_grad():
    for _ in range(3):
        m(xq, xkv, [])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        m(xq, xkv, [])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
print("PGRM fwd B=%d: %.3f ms  -> %.1f img/s/PGRM, %.1f TFLOP/s (1.1348 GFLOP/img)" % (B, dt * 1e3, B / dt, B * 1.1348e9 / dt / 1e12))

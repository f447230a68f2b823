#!/usr/bin/env python3
"""Phase timeline of k_ln_qkv_window_attn_bwd from a FAB_TIMING build (tools/build_variants.sh attn_fused_bwd "-DFAB_TIMING=1" fabtm):
DPMN_HIP_LIB=tools/variants/libdpmn_fabtm.so python tools/fab_timeline.py  -- s_memtime ticks (100 MHz constant clock -> 10 ns)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpmn_amd import ops, _abi
from dpmn_amd.utils import synth
B, H, W, C = 48, 16, 64, 96
dev = torch.device("cuda:0")
u = lambda n, s, lo=-1, hi=1: synth.uniform(n, s, lo, hi, 3).to(dev)
tq, tkv, dout = u("tq", (B, H * W, C)), u("tkv", (B, H * W, C)), u("do", (B, H * W, C))
ln = [u("a", (C,), .5, 1.5), u("b", (C,)), u("c", (C,), .5, 1.5), u("d", (C,))]
wq, bq, wkv, bkv = u("wq", (C, C), -.1, .1), u("bq", (C,)), u("wkv", (2 * C, C), -.1, .1), u("bkv", (2 * C,))
tables = [u("t%d" % i, ((2 * w - 1) ** 2, 2)) for i, w in enumerate((2, 4, 8))]
fold = []
ops.ln_qkv_window_attn_train(tq, tkv, *ln, wq, bq, wkv, bkv, tables, [2, 4, 8], [1, 2, 4], 2, H, W, save_qkv=False, fold_out=fold)
run = lambda: ops.ln_qkv_window_attn_bwd(tq, tkv, *ln, wq, bq, wkv, bkv, tables, [2, 4, 8], [1, 2, 4], 2, H, W, dout, fold=fold[0])
lib = ctypes.CDLL(_abi.LIB_PATH)
for _ in range(10):
    run()
torch.cuda.synchronize()
lib.dpmn_fab_timing_clear()
run()
torch.cuda.synchronize()
buf = np.zeros((512, 9, 8), dtype=np.uint64)
lib.dpmn_fab_timing_dump(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.astype(np.int64)
names = ["stats+proj", "sync(prev B)", "tiles->LDS+sync", "pass A", "sync", "pass B"]
T0 = min(int(x) for x in t.reshape(-1) if x > 0)
print("per block of XCD 0 (j = blockIdx / 8): units, [start, weights staged, loop end, flush end] in ticks (10 ns) since the first stamp")
for j in range(64):
    nu = int((t[8 * j][:8, 0] > 0).sum())
    print("  j=%2d units %d %s" % (j, nu, [int(x) - T0 if x else None for x in t[8 * j][8][:4]]))
for blk in (0, 8 * 10, 8 * 31, 8 * 33, 8 * 47, 8 * 50, 8 * 63):
    tb = t[blk]
    t0 = int(tb[8][0])
    print("block %d:" % blk)
    for uidx in range(8):
        r = tb[uidx]
        if r[0] == 0:
            continue
        d = [int(r[k + 1]) - int(r[k]) for k in range(6)]
        print("  unit %d start %6d: " % (uidx, int(r[0]) - t0) + "  ".join("%s %d" % (n, x) for n, x in zip(names, d)) + "   total %d" % (int(r[6]) - int(r[0])))

#!/usr/bin/env python3
"""Phase timeline of k_ln_qkv_window_attn from an FA_TIMING build (tools/build_variants.sh attn_fused "-DFA_TIMING=1" tm):
DPMN_HIP_LIB=tools/variants/libdpmn_tm.so python tools/fa_timeline.py  -- s_memtime ticks (100 MHz constant clock -> 10 ns)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpmn_amd import ops, _abi
from dpmn_amd.utils import synth
B, H, W, C = 48, 16, 64, 96
dev = torch.device("cuda:0")
u = lambda n, s, lo=-1, hi=1: synth.uniform(n, s, lo, hi, 3).to(dev)
tq, tkv = u("tq", (B, H * W, C)), u("tkv", (B, H * W, C))
ln = [u("a", (C,), .5, 1.5), u("b", (C,)), u("c", (C,), .5, 1.5), u("d", (C,))]
wq, bq, wkv, bkv = u("wq", (C, C), -.1, .1), u("bq", (C,)), u("wkv", (2 * C, C), -.1, .1), u("bkv", (2 * C,))
tables = [u("t%d" % i, ((2 * w - 1) ** 2, 2)) for i, w in enumerate((2, 4, 8))]
run = lambda: ops.ln_qkv_window_attn(tq, tkv, *ln, wq, bq, wkv, bkv, tables, [2, 4, 8], [1, 2, 4], 2, H, W)
for _ in range(10):
    run()
torch.cuda.synchronize()
_abi.lib.dpmn_fa_timing_clear()
run()
torch.cuda.synchronize()
buf = np.zeros((512, 9, 8), dtype=np.uint64)
_abi.lib.dpmn_fa_timing_dump(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.astype(np.int64)
names = ["stats", "proj", "fixup+prefetch", "barrier", "head0", "head1"]
T0 = min(int(x) for x in t.reshape(-1) if x > 0)
print("per block of XCD 0 (j = blockIdx / 8): [slot0 start, ready, slot1 start, ready, slot2 start, ready, end(not last slot), end(last slot)] in ticks since the first stamp")
for j in range(64):
    print("  j=%2d %s" % (j, [int(x) - T0 if x else None for x in t[8 * j][8]]))
for blk in (0, 8 * 20, 8 * 30, 8 * 45, 8 * 63):
    tb = t[blk]
    t0 = min(int(x) for x in tb.reshape(-1) if x > 0)
    print("block %d: slot stamps (ticks since first): %s" % (blk, [int(x) - t0 if x else None for x in tb[8]]))
    for uidx in range(8):
        r = tb[uidx]
        if r[0] == 0:
            continue
        d = [int(r[k + 1]) - int(r[k]) for k in range(6)]
        print("  unit %d start %6d: " % (uidx, int(r[0]) - t0) + "  ".join("%s %d" % (n, x) for n, x in zip(names, d)) + "   total %d" % (int(r[6]) - int(r[0])))

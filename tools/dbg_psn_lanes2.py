"""Debug: stages of the TATT PSN (head conv, text-prior interpreter, trunk) on two streams at once vs sequentially."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload, _abi
from dpmn_amd.utils import synth
dev = torch.device("cuda:0")
_abi.check(_abi.lib.dpmn_set_compute_dtype(int(os.environ.get("DBG_MODE", "2"))))
sr, models, psn, inp = workload.build("cfg1", batch=6)
bs = []
for i in range(4):
    b = synth.synth_batch(6, seed=60 + i)
    bs.append((b["images_lr"].to(dev), b["label_vecs"].to(dev)))
def stages(lr, lv):
    psn._check_mode()
    P = psn._trunk_pack()
    b1 = psn._head(lr, P)
    tp, prw = psn._tp_interpreter(b1, lv, P)
    out = psn._trunk(b1, P, tp)
    return {"b1": b1.clone(), "tp": tp.clone(), "out": out.clone()}
seq = [stages(*b) for b in bs]
torch.cuda.synchronize()
lanes = [torch.cuda.Stream(dev) for _ in range(2)]
cnt = {}
for rep in range(int(os.environ.get("DBG_REPS", "40"))):
    got = []
    for i, b in enumerate(bs):
        with torch.cuda.stream(lanes[i % 2]):
            got.append(stages(*b))
    torch.cuda.synchronize()
    for i in range(4):
        for k in ("b1", "tp", "out"):
            if not torch.equal(got[i][k], seq[i][k]):
                cnt[k] = cnt.get(k, 0) + 1
                if cnt[k] <= 2:
                    d = (got[i][k] - seq[i][k]).abs()
                    d = d.reshape(6, -1)
                    print("rep %d batch %d stage %s: max %.2e, images %s" % (rep, i, k, float(d.max()), torch.nonzero(d.amax(1) > 0).reshape(-1).tolist()))
print("mismatches per stage:", cnt)

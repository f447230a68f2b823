import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpmn_amd import workload, _abi
_abi.check(_abi.lib.dpmn_set_compute_dtype(2))
sr, models, psn, inp = workload.build("cfg1", batch=6)
for _ in range(2):
    out = psn(inp["images_lr"], inp["label_vecs"])
torch.cuda.synchronize()

"""Ordered kernel list of one CMM training forward + backward at the bench batch, from a rocprofv3 kernel trace.
  run:   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cmmtrace -- python tools/trace_cmm_train.py run
  read:  python tools/trace_cmm_train.py read gpurun_out/cmmtrace
(the tracer serialises the streams: durations are each kernel's own, the order is issue order per stream)"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    from dpmn_amd.model.cmm import ComplementationModulationModule as CMM
    from dpmn_amd.train.optim import Trainer
    dev = torch.device("cuda:0")
    m = CMM(c_img=3, cnum=64).to(dev).train()
    for p in m.parameters():
        p.requires_grad = True
    tr = Trainer([m], lr=1e-3, beta1=0.5, max_norm=0.25)
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    x1 = torch.rand(B, 3, 32, 128, device=dev).requires_grad_(True)
    x2 = torch.rand(B, 3, 32, 128, device=dev).requires_grad_(True)
    cot = torch.rand(B, 3, 32, 128, device=dev)
    for it in range(4):
        tr.zero_grad()
        out = m(x1, x2)
        torch.cuda.synchronize()
        (out * cot).sum().backward()
        torch.cuda.synchronize()
        tr.step()
        torch.cuda.synchronize()
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last iteration: from the last fill / zero kernel before the final forward -- simply take the last quarter by count
    n = len(rows) // 4
    rows = rows[-n:]
    t0 = int(rows[0]["Start_Timestamp"])
    tot = {}
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:60]
        print("%9.1f  %7.1f us  q%-3s %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r.get("Queue_Id", "?"), name, r.get("Grid_Size", "")))
        tot[name] = tot.get(name, 0) + d
    print()
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
        print("%8.1f us  %s" % (v, k))
    print("sum of kernel durations: %.2f ms" % (sum(tot.values()) / 1e3))

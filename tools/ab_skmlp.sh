#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/skmlp; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pgrm.py tests/test_gpu_train.py -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
for x in 0 1; do
  for r in 1 2; do
    DPMN_SKMLP=$x DPMN_SKMLP_TRAIN=$x timeout 400 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_${x}_$r.json
  done
done
python - <<'P'
import json
for x in (0,1):
    for r in (1,2):
        try:
            d=json.load(open("gpurun_out/skmlp/bench_%d_%d.json"%(x,r)))
            print("SKMLP",x,"run",r,"fwd",d["ms_per_step"],"train",d["train"]["ms_per_step"],"drop",d["train"]["with_dropout_0.1"]["ms_per_step"])
            print("   fwd", [(k["kernel"],k["launches_per_step"],k["us_per_launch"],k["frac"]) for k in d["kernels"] if "wstat" in k["kernel"] or "kloop" in k["kernel"]])
            print("   trn", [(k["kernel"],k["launches_per_step"],k["us_per_launch"],k["frac"]) for k in d["train"]["kernels"][:6]])
        except Exception as e: print("SKMLP",x,r,"failed",e)
P

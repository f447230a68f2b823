#!/bin/bash
# A/B of the in-launch split-K reduction (DPMN_CONV_XRED=0/1) on the MI355X box: tests, per-layer conv times, forward bench.
cd $GRAFT_REPO_ROOT
O=gpurun_out/xred; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_xred.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
for x in 0 1; do
  DPMN_CONV_XRED=$x timeout 300 python tools/prof_convs.py > $O/convs_$x.log 2>&1
  for r in 1 2; do
    DPMN_CONV_XRED=$x timeout 300 python bench.py --no-cpu-baseline --no-train --steps 30 --warmup 5 2>/dev/null | tail -1 > $O/bench_${x}_$r.json
  done
done
python - <<'P'
import json
for x in (0,1):
    for r in (1,2):
        try:
            d=json.load(open("gpurun_out/xred/bench_%d_%d.json"%(x,r)))
            print("XRED",x,"run",r,d["ms_per_step"],"ms",[ (k["kernel"],k["us_per_launch"],k["frac"]) for k in d["kernels"][:4]])
        except Exception as e: print("XRED",x,r,"failed",e)
P
paste -d'|' <(grep -E "^in|^convT" $O/convs_0.log | awk '{print $0}' | cut -c1-120) <(grep -E "^in|^convT" $O/convs_1.log | grep -oE "[0-9.]+ us +[0-9.]+ TF") | grep -E "cout +(128|256|512)" | head -40

#!/bin/bash
# Is the eager training step launch-bound?  The same step (a) eager on ONE stream, (b) replayed from one hipGraph capture (no CPU launch
# cost at all, one stream), (c) eager with the branch / weight-gradient streams (the product configuration).  Untraced wall-clock times.
cd $GRAFT_REPO_ROOT
O=gpurun_out/graph_vs_eager.txt; : > $O
run() { python bench.py --no-cpu-baseline --no-kernel-profile --mode train --steps 20 --warmup 6 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for r in 1 2; do
  echo "run $r: eager, one stream            $(DPMN_TRAIN_BRANCH_STREAMS=0 DPMN_WGRAD_STREAM=0 run) ms" >> $O
  echo "run $r: hipGraph replay, one stream  $(DPMN_TRAIN_BRANCH_STREAMS=0 DPMN_WGRAD_STREAM=0 run --graph) ms" >> $O
  echo "run $r: eager, three streams         $(run) ms" >> $O
done
cat $O

#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "768 32" "512 32" "512 16" "768 16" "1024 32" "384 16" "768 64"; do
  set -- $cfg
  echo -n "WG_BLOCKS=$1 CAP_MB=$2: "
  DPMN_WG_BLOCKS=$1 DPMN_WG_CAP_MB=$2 python bench.py --no-cpu-baseline --no-kernel-profile --mode train --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done

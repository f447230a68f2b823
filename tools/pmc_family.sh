#!/bin/bash
# HBM traffic of the forward bench's kernel families under an environment setting:  tools/pmc_family.sh <tag> [VAR=value ...]
# (two separate rocprofv3 --pmc passes, FETCH_SIZE and WRITE_SIZE; reduced by tools/collect_profiles.py --reduce)
set -u
R=$GRAFT_REPO_ROOT; tag=$1; shift
OUT=$R/gpurun_out/pmc_$tag; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do export "$v"; done
B="python $R/bench.py --no-cpu-baseline --no-kernel-profile --no-train --steps 3 --warmup 2"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
cd $R
python tools/collect_profiles.py --reduce $OUT > $OUT/reduce.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*counter_collection.csv" -delete 2>/dev/null
python tools/pmc_family.py $OUT $tag

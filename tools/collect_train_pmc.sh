#!/bin/bash
# only the three PMC passes over the training bench (tools/collect_profiles.sh runs them after the forward ones); adds to gpurun_out/prof
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; mkdir -p $OUT
rm -rf $OUT/pmc_fetch_train $OUT/pmc_write_train $OUT/pmc_mfma_train
cd /tmp && export TMPDIR=/tmp
T="python $R/bench.py --no-cpu-baseline --no-kernel-profile --no-train --mode train --steps 3 --warmup 2"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_train -- $T > $OUT/pmc_fetch_train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_train -- $T > $OUT/pmc_write_train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_train -- $T > $OUT/pmc_mfma_train.log 2>&1
cd $R
python tools/collect_profiles.py --reduce $OUT > $OUT/reduce.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*counter_collection.csv" -delete 2>/dev/null
tail -3 $OUT/reduce.log

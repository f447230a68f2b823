#!/bin/bash
# Evidence run on the MI355X box (gpurun):  bash tools/collect_profiles.sh [quick]
#   kernel-trace summaries of the forward and training benches, three PMC passes over the forward bench (FETCH_SIZE, WRITE_SIZE,
#   MFMA utilisation -- separate runs, --pmc never combined with other trace domains), and the bench JSON lines.
# Everything lands in gpurun_out/prof/; tools/collect_profiles.py <tag> copies the summaries into profiles/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
# the whole GPU suite first (both compute modes): its error record travels as <tag>_parity_errors.json
(cd $R && timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $OUT/gpu_suite.txt)
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-profile --no-train"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fwd -- $B --steps 10 --warmup 3 > $OUT/fwd.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- $B --mode train --steps 5 --warmup 2 > $OUT/train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 3 --warmup 2 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 3 --warmup 2 > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- $B --steps 3 --warmup 2 > $OUT/pmc_mfma.log 2>&1
# BASELINE.json's second metric: MFMA-busy over PGRM forwards only (bench.py pgrm_mfma_util reads the TOTAL row)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_pgrm -- python $R/tools/prof_pgrm.py 48 > $OUT/pmc_mfma_pgrm.log 2>&1
python $R/tools/pmc_pgrm_util.py $OUT/pmc_mfma_pgrm $OUT/pmc_pgrm_mfma_util.csv > /dev/null 2>&1
# mode 2 (f32 via bf16x3): the same forward / training benches, kernel-trace summaries + traffic + MFMA-busy (bf16 MOPS counted too)
X="$B --dtype x3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fwd_x3 -- $X --steps 10 --warmup 3 > $OUT/fwd_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_x3 -- $X --mode train --steps 5 --warmup 2 > $OUT/train_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_x3 -- $X --steps 3 --warmup 2 > $OUT/pmc_fetch_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_x3 -- $X --steps 3 --warmup 2 > $OUT/pmc_write_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_x3 -- $X --steps 3 --warmup 2 > $OUT/pmc_mfma_x3.log 2>&1
if [ "${1:-}" != "quick" ]; then
  T="$B --mode train --steps 3 --warmup 2"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_train -- $T > $OUT/pmc_fetch_train.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_train -- $T > $OUT/pmc_write_train.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_train -- $T > $OUT/pmc_mfma_train.log 2>&1
  TX="$X --mode train --steps 3 --warmup 2"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_train_x3 -- $TX > $OUT/pmc_fetch_train_x3.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_train_x3 -- $TX > $OUT/pmc_write_train_x3.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma_train_x3 -- $TX > $OUT/pmc_mfma_train_x3.log 2>&1
fi
cd $R
timeout 900 python bench.py 2>$OUT/bench_default.err | tail -1 > $OUT/bench_default.json
timeout 600 python bench.py --mode train 2>/dev/null | tail -1 > $OUT/bench_train.json
timeout 600 python bench.py --dtype x3 --no-cpu-baseline --no-train 2>/dev/null | tail -1 > $OUT/bench_x3.json
timeout 600 python bench.py --dtype x3 --mode train --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_train_x3.json
if [ "${1:-}" != "quick" ]; then
  timeout 600 python bench.py --mode train --drop 0 2>/dev/null | tail -1 > $OUT/bench_train_nodrop.json
  timeout 900 python bench.py --workload cfg4 --mode train --steps 5 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4_train.json
  timeout 600 python bench.py --workload cfg3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg3.json
  timeout 900 python bench.py --workload cfg4 --steps 5 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4.json
fi
if [ "${1:-}" != "quick" ]; then
  # the RCCL code path priced at world size 1 (ZeRO-1 reduce-scatter / all-gather, segmented CMM exchange), fp32 and mode 2
  for dt in f32 x3; do
    DPMN_FORCE_DIST=1 NCCL_DEBUG=VERSION timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --mode train --dtype $dt --no-cpu-baseline >> $OUT/rccl_world1.log 2>&1
  done
fi
if [ "${1:-}" != "quick" ]; then
  # where the cycles of the MFMA families go: SQ issue / stall / LDS counters of the forward, fp32 and mode 2
  bash tools/pmc_sq.sh f32 > /dev/null 2>&1
  bash tools/pmc_sq.sh x3 DPMN_COMPUTE_DTYPE=x3 BENCH_ARGS="--dtype x3" > /dev/null 2>&1
  cp gpurun_out/pmc_sq_f32.txt gpurun_out/pmc_sq_x3.txt $OUT/ 2>/dev/null
  # torch's own device ops per training step
  timeout 600 python tools/prof_torch_ops.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $OUT/torch_ops_per_step.txt
  timeout 900 python bench.py --workload cfg4 --steps 5 --warmup 6 --no-cpu-baseline --dtype x3 2>/dev/null | tail -1 > $OUT/bench_cfg4_x3.json
  timeout 900 python bench.py --workload cfg4 --mode train --steps 5 --warmup 3 --no-cpu-baseline --dtype x3 2>/dev/null | tail -1 > $OUT/bench_cfg4_train_x3.json
  timeout 600 python bench.py --workload cfg3 --no-cpu-baseline --dtype x3 2>/dev/null | tail -1 > $OUT/bench_cfg3_x3.json
fi
# keep the merge-back small: the per-dispatch traces are reduced on the box, only summaries travel
python tools/collect_profiles.py --reduce $OUT > $OUT/reduce.log 2>&1
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
find $OUT -name "*counter_collection.csv" -delete 2>/dev/null
find $OUT -type f -size +8M -delete 2>/dev/null
du -sh $OUT; cat $OUT/reduce.log | tail -5
head -c 400 $OUT/bench_default.json

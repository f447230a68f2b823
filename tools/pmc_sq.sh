#!/bin/bash
# SQ issue / stall counters per kernel of the forward bench:  tools/pmc_sq.sh <tag> [VAR=value ...]   (e.g. DPMN_COMPUTE_DTYPE=x3)
# Two rocprofv3 --pmc passes (8 SQ slots each, kernel trace only -- no other trace domains), reduced by tools/pmc_sq.py into
# gpurun_out/pmc_sq_<tag>.txt: per kernel the share of wave cycles parked (SQ_WAIT_ANY), issue-stalled (SQ_WAIT_INST_ANY, of which LDS),
# issuing (SQ_ACTIVE_INST_ANY), MFMA-busy, VALU / LDS instruction counts and LDS bank-conflict cycles.
set -u
R=$GRAFT_REPO_ROOT; tag=$1; shift
OUT=$R/gpurun_out/pmc_sq_$tag; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do export "$v"; done
B="python $R/bench.py --no-cpu-baseline --no-kernel-profile --no-train --steps 2 --warmup 2 --pipeline 1 ${BENCH_ARGS:-}"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1
cd $R
python tools/pmc_sq.py $OUT > $R/gpurun_out/pmc_sq_$tag.txt 2>&1
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*.csv" -size +20M -delete 2>/dev/null
tail -40 $R/gpurun_out/pmc_sq_$tag.txt

#!/bin/bash
# Round-end evidence run on the MI355X box (gpurun): kernel-trace summaries of the forward and training benches, the two
# PMC passes of the roofline kernel (separate runs, --pmc never combined with other trace domains), and the default bench
# line.  Everything lands in gpurun_out/final/; tools/final_profiles_collect.py copies the summaries into profiles/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fwd -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/fwd.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python $R/bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $OUT/train.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/tools/roofline_kernel.py 48 > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/tools/roofline_kernel.py 48 > $OUT/pmc_write.log 2>&1
cd $R
timeout 900 python bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log > $OUT/bench_default.json
timeout 600 python bench.py --mode train --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_train.json
timeout 600 python bench.py --mode train --drop 0.1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_train_drop.json
timeout 600 python bench.py --workload cfg3 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_cfg3.json
# keep the merge-back small: drop everything but the csv summaries
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -type f -size +20M -delete 2>/dev/null
du -sh $OUT
tail -c 600 $OUT/bench_default.json

mkdir -p gpurun_out/hp
DPMN_TEST_MODES=x3 timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_cmm.py tests/test_gpu_psn.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/hp/tests.txt
for r in 1 2; do
python bench.py --dtype x3 --no-cpu-baseline --no-train > gpurun_out/hp/f_$r.json 2>/dev/null
python bench.py --mode train --dtype x3 --no-cpu-baseline --steps 20 > gpurun_out/hp/t_$r.json 2>/dev/null
done

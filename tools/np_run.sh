mkdir -p gpurun_out/pp
for p in 2 3 4; do
python bench.py --pipeline $p --no-cpu-baseline --no-train --no-x3 --no-kernel-profile > gpurun_out/pp/f32_p$p.json 2>/dev/null
python bench.py --pipeline $p --no-cpu-baseline --no-train --dtype x3 --no-kernel-profile > gpurun_out/pp/x3_p$p.json 2>/dev/null
done

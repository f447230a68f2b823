# A/B of one environment switch over the forward and training benches: bash tools/ab.sh VAR "val1 val2 ..." [fwd|train|both]
VAR=$1; VALS=$2; WHAT=${3:-both}
for v in $VALS; do
  if [ "$WHAT" != train ]; then echo "== $VAR=$v fwd"; env $VAR=$v python bench.py --no-train --no-cpu-baseline --no-kernel-profile --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')"; fi
  if [ "$WHAT" != fwd ]; then echo "== $VAR=$v train"; env $VAR=$v python bench.py --mode train --no-cpu-baseline --no-kernel-profile --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'img/s')"; fi
done

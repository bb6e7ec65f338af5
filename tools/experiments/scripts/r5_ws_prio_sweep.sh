set -e
mkdir -p gpurun_out/prio
for rep in 1 2; do
for v in 0 1 2 4 5 6 8 9; do
  MI355_CONV_STAGGER=$v python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/prio/s${v}_$rep.json 2>gpurun_out/prio/err.log
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prio/s*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['avg_launch_us'])
PY

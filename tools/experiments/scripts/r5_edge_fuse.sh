set -e
mkdir -p gpurun_out/edge
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "edge_convs or euler or cfm or sampler" > gpurun_out/edge/tests.log 2>&1 || { tail -40 gpurun_out/edge/tests.log; exit 1; }
tail -3 gpurun_out/edge/tests.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/edge/on_$i.json 2>gpurun_out/edge/err.log
  MI355_CONV_EDGE=3 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/edge/off_$i.json 2>>gpurun_out/edge/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/edge/o*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['box']['launch_us'])
PY

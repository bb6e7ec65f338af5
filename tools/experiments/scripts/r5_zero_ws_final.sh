set -e
mkdir -p gpurun_out/zws
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/zws/def_$i.json 2>gpurun_out/zws/err.log
  MI355_CONV_PP=61 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/zws/zero_$i.json 2>>gpurun_out/zws/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/zws/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; print(f, d['value'], r['box']['launch_us'], r['kernel'][:40], r['frac'], r['avg_launch_us'])
PY

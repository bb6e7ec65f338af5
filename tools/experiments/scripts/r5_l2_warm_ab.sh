set -e
mkdir -p gpurun_out/warm
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/warm/w1_$i.json 2>gpurun_out/warm/err.log
  MI355_L2_WARM=0 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/warm/w0_$i.json 2>>gpurun_out/warm/err.log
done
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-out gpurun_out/warm/per_op_w1.json > /dev/null 2>>gpurun_out/warm/err.log
MI355_L2_WARM=0 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-out gpurun_out/warm/per_op_w0.json > /dev/null 2>>gpurun_out/warm/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/warm/w*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['box']['launch_us'])
PY

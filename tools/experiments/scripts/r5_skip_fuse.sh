set -e
mkdir -p gpurun_out/skf
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "skip_connection_rides or concat_groupnorm or small_conv_epilogue" > gpurun_out/skf/tests.log 2>&1 || { tail -40 gpurun_out/skf/tests.log; exit 1; }
tail -3 gpurun_out/skf/tests.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/skf/on_$i.json 2>gpurun_out/skf/err.log
  MI355_CONV_SMALL=7 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/skf/off_$i.json 2>>gpurun_out/skf/err.log
done
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-out gpurun_out/skf/per_op.json > /dev/null 2>>gpurun_out/skf/err.log
MI355_CONV_SMALL=7 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-out gpurun_out/skf/per_op_off.json > /dev/null 2>>gpurun_out/skf/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/skf/o*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['box']['launch_us'])
PY

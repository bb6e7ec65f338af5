set -e
mkdir -p gpurun_out/afin
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "attention_norm_finalized or attention_block or cfg2 or unet_forward_bf16 or attn" > gpurun_out/afin/tests.log 2>&1 || { tail -40 gpurun_out/afin/tests.log; exit 1; }
tail -3 gpurun_out/afin/tests.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/afin/on_$i.json 2>gpurun_out/afin/err.log
  MI355_GN_EPILOGUE=7 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/afin/off_$i.json 2>>gpurun_out/afin/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/afin/o*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['box']['launch_us'], d['roofline']['other_kernels']['attention_block']['avg_launch_us'])
PY

set -e
mkdir -p gpurun_out/pk
python -m pytest tests -m gpu -x -q -k "F16 or f16 or fp16" > gpurun_out/pk/tests.log 2>&1 || { tail -30 gpurun_out/pk/tests.log; exit 1; }
tail -3 gpurun_out/pk/tests.log
V=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc/libmi355_sampler_a0_p0_nopk.so
for i in 1 2 3; do
  python bench.py --precision fp16 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pk/pk_$i.json 2>gpurun_out/pk/err.log
  MI355_SAMPLER_LIB=$V python bench.py --precision fp16 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pk/nopk_$i.json 2>>gpurun_out/pk/err.log
done
python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pk/bf16.json 2>>gpurun_out/pk/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/pk/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline'].get('box'))
PY
python tools/quality_delta.py --x2 > gpurun_out/pk/quality.log 2>&1; tail -12 gpurun_out/pk/quality.log

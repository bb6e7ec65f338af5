set -e
mkdir -p gpurun_out/cgn
python -m pytest tests/test_gpu_unet.py -m gpu -x -q -k "concat_groupnorm or small_conv_epilogue or in_place_at_16" > gpurun_out/cgn/tests.log 2>&1 || { tail -40 gpurun_out/cgn/tests.log; exit 1; }
tail -3 gpurun_out/cgn/tests.log
H=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc/libmi355_sampler_head.so
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/cgn/on_$i.json 2>gpurun_out/cgn/err.log
  MI355_GN_EPILOGUE=3 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/cgn/off_$i.json 2>>gpurun_out/cgn/err.log
  MI355_SAMPLER_LIB=$H python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/cgn/head_$i.json 2>>gpurun_out/cgn/err.log
done
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-out gpurun_out/cgn/per_op.json > /dev/null 2>>gpurun_out/cgn/err.log
MI355_SAMPLER_LIB=$H python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-out gpurun_out/cgn/per_op_head.json > /dev/null 2>>gpurun_out/cgn/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/cgn/[oh]*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['box']['launch_us'])
PY

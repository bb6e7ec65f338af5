set -e
mkdir -p gpurun_out/pp1h
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "pingpong_conv1x1 or cfg2 or groupnorm_in_place or unet_forward_bf16 or attention_block" > gpurun_out/pp1h/tests.log 2>&1 || { tail -40 gpurun_out/pp1h/tests.log; exit 1; }
tail -3 gpurun_out/pp1h/tests.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pp1h/on_$i.json 2>gpurun_out/pp1h/err.log
  MI355_CONV_PP=13 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pp1h/off_$i.json 2>>gpurun_out/pp1h/err.log
done
python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-out gpurun_out/pp1h/per_op.json > /dev/null 2>>gpurun_out/pp1h/err.log
MI355_CONV_PP=13 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --profile-out gpurun_out/pp1h/per_op_off.json > /dev/null 2>>gpurun_out/pp1h/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/pp1h/o*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['box']['launch_us'])
d=json.load(open('gpurun_out/pp1h/per_op.json'))['ops']; o=json.load(open('gpurun_out/pp1h/per_op_off.json'))['ops']
for i,(x,y) in enumerate(zip(d,o)):
    if x['kind']=='conv' and x['ks']==1 and x['h']==16: print(i,x['cin'],x['cout'], round(x['ms']*1000,1), round(y['ms']*1000,1))
PY

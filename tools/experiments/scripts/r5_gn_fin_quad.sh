set -e
mkdir -p gpurun_out/finq
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "cfg2 or unet_forward or groupnorm or film or gn_" > gpurun_out/finq/tests.log 2>&1 || { tail -40 gpurun_out/finq/tests.log; exit 1; }
tail -3 gpurun_out/finq/tests.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/finq/on_$i.json 2>gpurun_out/finq/err.log
  MI355_GN_FIN_QUAD=0 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/finq/off_$i.json 2>>gpurun_out/finq/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/finq/o*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['box']['launch_us'])
PY
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/finq/st -o s -- python3 $R/bench.py --steps 1 --warmup 1 --nfe 10 --no-cpu-baseline > $R/gpurun_out/finq/stats.log 2>&1
cd $R && python tools/rocpd_stats.py gpurun_out/finq/st/s_results.db gpurun_out/finq/stats.csv && grep -i "gn_finalize" gpurun_out/finq/stats.csv | cut -c1-200
rm -rf gpurun_out/finq/st

#!/bin/bash
# round 5: the trimmed prologue (prescaled exponent tables, pair math): parity, isolated timings vs ws, e2e A/B
O=gpurun_out/${TAG:-r5_pp_trim}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "pingpong" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -3 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
export MI355_CONV_TIME=100
{
for rep in 1 2; do
  for shape in "256 128 32 128 3" "256 256 32 128 3" "256 384 32 128 3" "256 256 16 256 3" "256 512 16 256 3"; do
    echo -n "shape $shape ws: "; MI355_CONV_PP=1 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
    echo -n "shape $shape pp: "; MI355_CONV_PP=13 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
  done
done
} 2>&1 | tee $O/times.txt
unset MI355_CONV_TIME
for rep in 1 2 3; do for pp in 1 5 13; do echo -n "pp$pp: "; MI355_CONV_PP=$pp python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'img/s')"; done; done 2>&1 | tee $O/bench_ab.txt

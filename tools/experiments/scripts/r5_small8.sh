#!/bin/bash
# round 5: 8x8-level conv with eight waves of 32 channels (conv_small bit 1) vs four waves of 64: parity, isolated timings, e2e
O=gpurun_out/${TAG:-r5_small8}; mkdir -p $O
MI355_CONV_SMALL=3 timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_unet.py -x -q -m gpu -k "ws_conv or small or cfg2_b256 or epilogue" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -3 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
export MI355_CONV_TIME=100
{
for rep in 1 2; do
  for shape in "256 256 8 256 3 nogn" "256 512 8 256 3 nogn"; do
    for sm in 1 3; do echo -n "shape $shape small=$sm: "; MI355_CONV_SMALL=$sm timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail; done
  done
done
} 2>&1 | tee $O/times.txt
unset MI355_CONV_TIME
for rep in 1 2 3; do for sm in 1 3; do echo -n "small=$sm: "; MI355_CONV_SMALL=$sm python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'img/s')"; done; done 2>&1 | tee $O/bench_ab.txt

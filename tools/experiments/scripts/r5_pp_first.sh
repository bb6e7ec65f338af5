#!/bin/bash
# round 5: first run of the prologue / narrow forms of the ping-pong conv: parity tests, then isolated same-box timings old (conv_pp=1: round-4 behaviour) vs new (13)
O=gpurun_out/${TAG:-r5_pp_first}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "pingpong" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -5 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
export MI355_CONV_TIME=100
{
for rep in 1 2; do
  for shape in "256 128 32 128 3" "256 256 32 128 3" "256 384 32 128 3" "256 128 16 256 3" "256 256 16 256 3" "256 384 16 256 3" "256 512 16 256 3" "256 128 32 128 3 nogn" "256 256 16 256 3 nogn"; do
    for pp in 1 13; do
      echo -n "shape $shape pp=$pp: "; MI355_CONV_PP=$pp timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" || echo fail
    done
  done
done
} 2>&1 | tee $O/times.txt

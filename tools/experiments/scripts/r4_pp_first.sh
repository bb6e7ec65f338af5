#!/bin/bash
# round 4: first GPU contact of the ping-pong conv: parity tests, then isolated launch times old / new on the same box
O=gpurun_out/${TAG:-r4_pp1}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -k "pingpong" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -5 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
run() { echo -n "$SHAPE | $1: "; shift; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE nogn 2>&1 | grep -E "conv time" | tail -1 | sed 's/\[conv time\] 30 launches, //'; }
{
for SHAPE in "256 256 16 256 3" "256 512 16 256 3" "256 128 16 256 3" "256 384 16 256 3" "512 256 16 256 3"; do
  for rep in 1 2; do
    run "ws " MI355_CONV_PP=0
    run "pp " MI355_CONV_PP=2
  done
done
} 2>&1 | tee $O/times.txt

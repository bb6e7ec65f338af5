#!/bin/bash
# round 4: 1x1 ping-pong conv: parity tests, then isolated launch times old (conv_pp = 0) vs new (conv_pp = 2) on one box
O=gpurun_out/${TAG:-r4_pp1x1}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -k "pingpong_conv1x1" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -4 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
run() { echo -n "$SHAPE | $1: "; shift; env "$@" MI355_CONV_TIME=100 timeout -k 10 120 python tools/time_conv.py $SHAPE nogn 2>&1 | grep -E "conv time" | tail -1 | sed 's/\[conv time\] 100 launches, //'; }
IFS=";" read -ra SH <<< "${SHAPES:-256 512 16 256 1;256 384 16 256 1;256 256 16 256 1;256 128 16 256 1;512 512 16 256 1}"; unset IFS
{
for SHAPE in "${SH[@]}"; do
  for rep in 1 2; do
    run "old" MI355_CONV_PP=0
    run "pp1" MI355_CONV_PP=2
  done
done
} 2>&1 | tee $O/times.txt

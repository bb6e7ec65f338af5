#!/bin/bash
# round 4: the streaming out-conv kernel: parity test, then isolated launch times old (conv_edge = 0) vs new on one box
O=gpurun_out/${TAG:-r4_edge}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "out_conv or conv2d" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -4 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
run() { echo -n "$SHAPE | $1: "; shift; env "$@" MI355_CONV_TIME=100 timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep -E "conv time" | tail -1 | sed 's/\[conv time\] 100 launches, //'; }
IFS=";" read -ra SH <<< "${SHAPES:-256 128 32 3 3;128 128 128 3 3;256 128 64 3 3}"; unset IFS
{
for SHAPE in "${SH[@]}"; do
  for rep in 1 2; do
    run "old " MI355_CONV_EDGE=0
    run "edge" MI355_CONV_EDGE=1
  done
done
} 2>&1 | tee $O/times.txt

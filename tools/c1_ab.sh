#!/bin/bash
# A/B on one box: 1x1 convs over wide inputs with one-chunk weight groups (MI355_CONV1X1_GC1=1) vs the previous choice (=0)
run() { echo -n "$1: "; shift; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE nogn 2>&1 | grep "conv time" | tail -1; }
for SHAPE in "256 512 16 256 1" "256 384 16 256 1" "256 384 32 128 1" "256 512 8 256 1" "256 256 32 128 1"; do
  echo "== $SHAPE"
  run "gc1=0" MI355_CONV1X1_GC1=0
  run "gc1=1" MI355_CONV1X1_GC1=1
  run "gc1=0" MI355_CONV1X1_GC1=0
  run "gc1=1" MI355_CONV1X1_GC1=1
done
for m in 1 0 1 0; do MI355_CONV1X1_GC1=$m python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/gc1=$m /"; done

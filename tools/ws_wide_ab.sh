#!/bin/bash
# A/B on one box: 256-channel tiles (MI355_CONV_WS256=1) vs 128-channel tiles (=0) of the persistent conv, isolated launches + bench
run() { echo -n "$1: "; shift; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep "conv time" | tail -1; }
for SHAPE in "256 256 16 256 3" "256 512 16 256 3" "256 256 32 256 3" "256 384 16 256 3" "256 128 16 256 3"; do
  echo "== $SHAPE"
  run "128-ch tiles" MI355_CONV_WS256=0
  run "256-ch tiles" MI355_CONV_WS256=1
  run "128-ch tiles" MI355_CONV_WS256=0
  run "256-ch tiles" MI355_CONV_WS256=1
done
for m in 1 0 1 0; do MI355_CONV_WS256=$m python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/ws256=$m /"; done

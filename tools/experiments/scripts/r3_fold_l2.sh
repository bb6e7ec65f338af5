#!/bin/bash
# is the folded 1x1 conv bound by where its pixels come from?  MI355_CONV_ABLATE=64: every tile's folded rows read the same 16 pixels (L2 hits)
O=gpurun_out/r3_fold_l2; mkdir -p $O
run() { local name=$1; shift; echo -n "$SHAPE $EXTRA | $name: "; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE $EXTRA 2>&1 | grep -E "conv time" | tail -1 | sed 's/\[conv time\] 30 launches, //'; }
{
for cfg in "256 128 32 128 3|" "256 128 32 128 3|fold=384" "256 256 16 256 3|" "256 256 16 256 3|fold=512"; do
  SHAPE=${cfg%%|*}; EXTRA=${cfg##*|}
  run "as is" X=1
  run "folded rows from L2" MI355_CONV_ABLATE=64
done
} 2>&1 | tee $O/l2.txt

#!/bin/bash
# isolated launches of the second ResBlock conv with / without the folded 1x1 skip conv, times + in-kernel stamps
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r3_fold_stamps}; mkdir -p $O
run() { local name=$1; shift; echo -n "$SHAPE $EXTRA | $name: "; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE $EXTRA 2>&1 | grep -E "conv time|conv stamps" | tail -1 | sed 's/\[conv time\] 30 launches, //'; }
{
for cfg in "256 128 32 128 3|" "256 128 32 128 3|fold=384" "256 128 32 128 3|fold=256" "256 256 16 256 3|" "256 256 16 256 3|fold=512" "256 256 16 256 3|fold=128" "256 128 32 128 3|nogn" "256 128 32 128 3|nogn fold=384"; do
  SHAPE=${cfg%%|*}; EXTRA=${cfg##*|}
  run "time" X=1
  run "time" X=1
  run "stamps" MI355_SAMPLER_LIB=$D/libmi355_sampler_stamps.so
done
} 2>&1 | tee $O/fold.txt

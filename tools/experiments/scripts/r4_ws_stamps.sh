#!/bin/bash
# round 4: in-kernel stamps of the warp-specialised conv WITH its GroupNorm prologue at K = 1152 vs 2304 (is there a per-tile bubble?)
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r4_ws_stamps}; mkdir -p $O
for SHAPE in "256 128 32 128 3" "256 256 32 128 3" "256 128 32 128 3 nogn"; do
echo "== $SHAPE"; MI355_CONV_PP=0 MI355_CONV_TIME=${REPS:-100} MI355_SAMPLER_LIB=$D/libmi355_sampler_stamps.so timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep -E "conv stamps|conv clock|conv time"
echo "-- timed build"; MI355_CONV_PP=0 MI355_CONV_TIME=${REPS:-100} timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep -E "conv time"
done 2>&1 | tee $O/stamps.txt

#!/bin/bash
# round 5: timeline of the ping-pong conv WITH the in-LDS prologue (stamps build with -DPP_TRACE_TAP0=12: taps 3 .. 6 of chunk 1; the transform taps of the wide form are 3, 5, 7)
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r5_pp_stamps}; mkdir -p $O
for GN in "" "nogn"; do
echo "== 256 256 16 256 3 $GN"; MI355_CONV_PP=14 MI355_CONV_TIME=${REPS:-200} MI355_SAMPLER_LIB=$D/libmi355_sampler_stamps.so timeout -k 10 120 python tools/time_conv.py 256 256 16 256 3 $GN 2>&1 | grep -E "conv stamps|conv clock|conv time|pp trace|L:|M:|E:|tile"
done 2>&1 | tee $O/stamps.txt

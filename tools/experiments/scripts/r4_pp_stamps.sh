#!/bin/bash
# round 4: in-kernel stamps + clock of the ping-pong conv (csrc: make stamps)
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r4_pp_stamps}; mkdir -p $O
IFS=";" read -ra SH <<< "${SHAPES:-256 256 16 256 3;256 512 16 256 3;512 256 16 256 3;256 128 16 256 3}"; unset IFS
for SHAPE in "${SH[@]}"; do
echo "== $SHAPE"; MI355_CONV_PP=2 MI355_CONV_TIME=${REPS:-200} MI355_SAMPLER_LIB=$D/libmi355_sampler_stamps.so timeout -k 10 120 python tools/time_conv.py $SHAPE nogn 2>&1 | grep -E "conv stamps|conv clock|conv time|pp trace"
done 2>&1 | tee $O/stamps.txt

#!/bin/bash
# in-kernel phase stamps of the persistent conv (csrc: make stamps), with / without prologue
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
mkdir -p gpurun_out/r3_stamps
for SHAPE in "256 128 32 128 3" "256 256 16 256 3" "256 512 16 256 3"; do for GNV in "" nogn; do
echo -n "$SHAPE $GNV: "; MI355_SAMPLER_LIB=$D/libmi355_sampler_stamps.so timeout -k 10 120 python tools/time_conv.py $SHAPE $GNV 2>&1 | grep "conv stamps" | tail -1
done; done | tee gpurun_out/r3_stamps/${TAG:-stamps}.txt

#!/bin/bash
# round 3, experiment 3: the consumer waves of the persistent conv alone (no loaders, no hand-over): what the MFMA + LDS-read + epilogue stream sustains
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/r3_exp3; mkdir -p $O
run() { local name=$1; shift; echo -n "$SHAPE $GNV | $name: "; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE $GNV 2>&1 | grep -E "conv time|conv stamps" | tail -1 | sed 's/\[conv time\] 30 launches, //'; }
for SHAPE in "256 128 32 128 3" "256 256 16 256 3" "256 512 16 256 3"; do
 GNV=nogn
  run "base (DMA loaders)" X=1
  run "base, no stores" MI355_CONV_ABLATE=1
  run "consumers alone" MI355_SAMPLER_LIB=$D/libmi355_sampler_a2048.so
  run "consumers alone, no stores" MI355_SAMPLER_LIB=$D/libmi355_sampler_a2048.so MI355_CONV_ABLATE=1
  run "consumers alone, no LDS reads" MI355_SAMPLER_LIB=$D/libmi355_sampler_a2560.so
  run "consumers alone, no LDS reads, no stores" MI355_SAMPLER_LIB=$D/libmi355_sampler_a2560.so MI355_CONV_ABLATE=1
done 2>&1 | tee $O/consumers_alone.txt

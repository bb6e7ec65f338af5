#!/bin/bash
# round 3, experiment 2: ablation matrix of the barrier-free persistent conv, with and without the GN+SiLU prologue
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/r3_exp2; mkdir -p $O
run() { local name=$1; shift; echo -n "$SHAPE $GNV | $name: "; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE $GNV 2>&1 | grep -E "conv time|conv stamps" | tail -1 | sed 's/\[conv time\] 30 launches, //'; }
for SHAPE in "256 128 32 128 3" "256 256 16 256 3"; do
 for GNV in "" "nogn"; do
  run "base" X=1
  run "no stores" MI355_CONV_ABLATE=1
  for v in 4 8 16 24 20 256 512; do run "ABL=$v" MI355_SAMPLER_LIB=$D/libmi355_sampler_a$v.so; done
  run "ABL=256 no stores" MI355_SAMPLER_LIB=$D/libmi355_sampler_a256.so MI355_CONV_ABLATE=1
  run "stamps" MI355_SAMPLER_LIB=$D/libmi355_sampler_stamps.so
 done
done 2>&1 | tee $O/ablations.txt

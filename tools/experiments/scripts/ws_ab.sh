#!/bin/bash
# A/B of two builds of the persistent conv on the same box: libmi355_sampler.so vs libmi355_sampler_old.so (isolated launches + bench)
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
run() { echo -n "$1: "; shift; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep "conv time" | tail -1; }
for SHAPE in "256 128 32 128 3" "256 256 16 256 3" "256 512 16 256 3" "256 256 32 128 3"; do
  echo "== $SHAPE"
  run "old" MI355_SAMPLER_LIB=$D/libmi355_sampler_old.so
  run "new" MI355_SAMPLER_LIB=$D/libmi355_sampler.so
  run "old" MI355_SAMPLER_LIB=$D/libmi355_sampler_old.so
  run "new" MI355_SAMPLER_LIB=$D/libmi355_sampler.so
done
for l in libmi355_sampler_old.so libmi355_sampler.so libmi355_sampler_old.so libmi355_sampler.so; do MI355_SAMPLER_LIB=$D/$l python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/$l /"; done

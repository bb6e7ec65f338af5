#!/bin/bash
# same-box A/B of two builds: libmi355_sampler_old.so vs libmi355_sampler.so (isolated conv launches, with / without prologue, + bench)
#   TAG=<dir under gpurun_out>  SHAPES="B Cin H Cout k;..."  NOBENCH=1  NOCONV=1
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r3_ab}; mkdir -p $O
run() { local name=$1; shift; echo -n "$SHAPE $GNV | $name: "; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE $GNV 2>&1 | grep -E "conv time" | tail -1 | sed 's/\[conv time\] 30 launches, //'; }
IFS=";" read -ra SH <<< "${SHAPES:-256 128 32 128 3;256 256 16 256 3;256 512 16 256 3;256 256 32 128 3}"; unset IFS
{
if [ -z "$NOCONV" ]; then
for SHAPE in "${SH[@]}"; do
 for GNV in "" "nogn"; do
  for rep in 1 2; do
   run "old" MI355_SAMPLER_LIB=$D/libmi355_sampler_old.so
   run "new" MI355_SAMPLER_LIB=$D/libmi355_sampler.so
  done
 done
done
fi
if [ -z "$NOBENCH" ]; then
for l in libmi355_sampler_old.so libmi355_sampler.so libmi355_sampler_old.so libmi355_sampler.so libmi355_sampler_old.so libmi355_sampler.so; do MI355_SAMPLER_LIB=$D/$l python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/$l /"; done
fi
} 2>&1 | tee $O/ab.txt

#!/bin/bash
# round 4: same-box interleaved A/B of library variants on isolated prologue-free conv launches
#   LIBS="name=path;..." (path relative to csrc/), SHAPES="B Cin H Cout k;...", TAG
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r4_pp_ab}; mkdir -p $O
IFS=";" read -ra SH <<< "${SHAPES:-256 256 16 256 3;256 512 16 256 3;256 384 16 256 3;256 128 16 256 3;512 256 16 256 3}"; unset IFS
IFS=";" read -ra LB <<< "${LIBS:-ws=libmi355_sampler.so@0;col=libmi355_sampler.so@2}"; unset IFS
{
for SHAPE in "${SH[@]}"; do
  for rep in 1 2 3; do
    for L in "${LB[@]}"; do
      name=${L%%=*}; rest=${L#*=}; lib=${rest%@*}; pp=${rest#*@}
      echo -n "$SHAPE | $name: "; MI355_SAMPLER_LIB=$D/$lib MI355_CONV_PP=$pp MI355_CONV_TIME=${REPS:-100} timeout -k 10 120 python tools/time_conv.py $SHAPE nogn 2>&1 | grep -E "conv time" | tail -1 | sed 's/\[conv time\] [0-9]* launches, //'
    done
  done
done
} 2>&1 | tee $O/ab.txt

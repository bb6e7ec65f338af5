#!/bin/bash
# round 5: where the in-LDS prologue's arithmetic runs (PP_TRFORM 0 / 1 / 2) - isolated same-box timings, interleaved, x2; ws = the round-4 path (conv_pp=1)
O=gpurun_out/${TAG:-r5_pp_forms}; mkdir -p $O
C=image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
export MI355_CONV_TIME=100
{
for rep in 1 2; do
  for shape in "256 128 32 128 3" "256 384 32 128 3" "256 256 16 256 3" "256 512 16 256 3"; do
    echo -n "shape $shape ws: "; MI355_CONV_PP=1 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
    for v in ${VARIANTS:-tf0 tf1 tf2 nomath}; do
      lib=$C/libmi355_sampler_a0_p0_$v.so; [ $v = nomath ] && lib=$C/libmi355_sampler_a0_p256_nomath.so
      echo -n "shape $shape $v: "; MI355_SAMPLER_LIB=$PWD/$lib MI355_CONV_PP=13 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
    done
  done
done
} 2>&1 | tee $O/times.txt

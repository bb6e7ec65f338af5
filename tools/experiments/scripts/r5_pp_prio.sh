#!/bin/bash
# round 5: wave priorities around the in-LDS prologue: m0 = no s_setprio in the M segment, t2 = priority 2 while the prologue's arithmetic runs; w = whole pieces, u = half-piece units
O=gpurun_out/${TAG:-r5_pp_prio}; mkdir -p $O
C=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
export MI355_CONV_TIME=100
{
for rep in 1 2; do
  for shape in "256 128 32 128 3" "256 256 16 256 3" "256 512 16 256 3" "256 256 16 256 3 nogn"; do
    for v in ${VARIANTS:-whole m0w t2w m0u t2u nomath}; do
      lib=$C/libmi355_sampler_a0_p0_$v.so; [ $v = nomath ] && lib=$C/libmi355_sampler_a0_p256_nomath.so
      echo -n "shape $shape $v: "; MI355_SAMPLER_LIB=$lib MI355_CONV_PP=13 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
    done
  done
done
} 2>&1 | tee $O/times.txt

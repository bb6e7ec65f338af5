#!/bin/bash
# round 4: per-op time of the fused attention block under compile-time ablations of the persistent kernel (make variant_src ... -DATTN_ABLATE=n)
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r4_attn_ablate}; mkdir -p $O
{
for a in ${ABLS:-0 1 2 3 4 8 16 24 32 64 0}; do
  lib=$D/libmi355_sampler_attn$a.so; [ "$a" = 0 ] && lib=$D/libmi355_sampler.so
  MI355_SAMPLER_LIB=$lib timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-out $O/p$a.json > /dev/null 2>&1
  echo -n "ablate=$a: "; python tools/show_profile.py $O/p$a.json | grep "attention c256"
done
} 2>&1 | tee $O/ablate.txt

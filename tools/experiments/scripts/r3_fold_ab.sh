#!/bin/bash
# round 3: ResBlock skip 1x1 conv folded into the second 3x3 conv.  Same box, interleaved:
#   nofold lib  = the library built with -DWS_FOLD=0 (no support code in the persistent conv): what the support costs plain launches
#   MI355_CONV_FOLD=0 / 1 = the shipped library with the fold off / on
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r3_fold}; mkdir -p $O
{
for rep in 1 2 3; do
 for v in "MI355_SAMPLER_LIB=$D/libmi355_sampler_nofold.so" "MI355_CONV_FOLD=0" "MI355_CONV_FOLD=1"; do
  echo -n "${v##*/}: "; env $v python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110
 done
done
MI355_CONV_FOLD=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-out $O/prof_fold.json > /dev/null 2>&1
MI355_CONV_FOLD=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-out $O/prof_nofold.json > /dev/null 2>&1
} 2>&1 | tee $O/ab.txt

#!/bin/bash
# same-box, interleaved A/B of several builds of the library: LIBS="head nofold ''" (suffixes of libmi355_sampler<_x>.so)
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/${TAG:-r3_libs}; mkdir -p $O
{
for rep in 1 2 3; do
 for l in ${LIBS:-head nofold main}; do
  f=$D/libmi355_sampler_$l.so; [ "$l" = main ] && f=$D/libmi355_sampler.so
  echo -n "$l: "; MI355_SAMPLER_LIB=$f python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110
 done
done
} 2>&1 | tee $O/ab.txt

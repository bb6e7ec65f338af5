#!/bin/bash
# persistent conv: back-off of the counter polls (WS_LSLEEP: loaders' buffer-free wait; WS_CSLEEP: consumers' hand-over), same box, interleaved
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/r3_sleep; mkdir -p $O
for i in 1 2 3; do
  for v in "" _c2 _c4; do
    MI355_SAMPLER_LIB=$D/libmi355_sampler$v.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-100 | sed "s/^/base$v /"
  done
done | tee $O/ab.txt

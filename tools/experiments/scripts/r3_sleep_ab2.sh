#!/bin/bash
# poll back-off (loaders 8 / consumers 2 vs 1 / 1) on the larger-image workloads, same box, interleaved
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/r3_sleep2; mkdir -p $O
for w in flowers64_superres_euler100_b256 px128_inpaint_ddim100_b128 cifar64_cfm_euler50_b256; do
for i in 1 2; do
  for v in _s11 ""; do
    MI355_SAMPLER_LIB=$D/libmi355_sampler$v.so python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'sleep1_1' if '$v' else 'sleep8_2', d['value'])"
  done
done
done | tee $O/ab.txt

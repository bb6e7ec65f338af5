#!/bin/bash
# in-kernel GroupNorm finalize (ws_gn_finalize): parity subset, then same-box A/B: HEAD library / new library with the launch / new library in-kernel
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/r3_fin_ab; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
grep -q " passed" $O/tests.txt || exit 1
for i in 1 2 3; do
  MI355_SAMPLER_LIB=$D/libmi355_sampler_old.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/head    /"
  MI355_GN_FIN_INLINE=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/launch  /"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/inline  /"
done | tee $O/ab.txt

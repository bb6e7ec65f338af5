#!/bin/bash
# 16x16 level: GroupNorm applied in place by the producing persistent conv: parity subset, then same-box interleaved A/B against the previous library
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/r3_wsact; mkdir -p $O
if [ -z "$SKIPTESTS" ]; then
timeout -k 10 700 python -m pytest tests/test_gpu_unet.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
grep -q " passed" $O/tests.txt || exit 1
grep -q "failed" $O/tests.txt && exit 1
fi
for i in 1 2 3; do
  MI355_SAMPLER_LIB=$D/libmi355_sampler_old.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-100 | sed "s/^/prev        /"
  MI355_SAMPLER_LIB=$D/libmi355_sampler_old.so MI355_GN_EPILOGUE=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-100 | sed "s/^/prev  noepi /"
  MI355_GN_EPILOGUE=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-100 | sed "s/^/new   noepi /"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-100 | sed "s/^/new         /"
done | tee $O/ab.txt

#!/bin/bash
# GroupNorm of the small levels in the producing conv's epilogue (MI355_GN_EPILOGUE): parity subset, then same-box interleaved A/B + per-op profiles
O=gpurun_out/r3_gnepi; mkdir -p $O
[ -n "$SKIPTESTS" ] || timeout -k 10 700 python -m pytest tests/test_gpu_unet.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
[ -n "$SKIPTESTS" ] || grep -q " passed" $O/tests.txt || exit 1
[ -z "$SKIPTESTS" ] && grep -q "failed" $O/tests.txt && exit 1
for i in 1 2 3; do
  MI355_GN_EPILOGUE=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/pass     /"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/epilogue /"
done | tee $O/ab.txt
MI355_GN_EPILOGUE=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-out $O/prof_pass.json > /dev/null 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-out $O/prof_epi.json > /dev/null 2>&1

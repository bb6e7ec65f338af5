#!/bin/bash
# persistent conv: image-major tile walk vs the XCD-interleaved one (MI355_CONV_ABLATE=128), same box, interleaved
O=gpurun_out/r3_walk; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "ws or persistent or cfg2" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for i in 1 2 3; do
  MI355_CONV_ABLATE=128 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/xcd-walk /"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-110 | sed "s/^/img-walk /"
done | tee $O/ab.txt

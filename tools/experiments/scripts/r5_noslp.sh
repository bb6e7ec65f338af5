#!/bin/bash
# round 5: -fno-slp-vectorize on conv_igemm.hip (the in-LDS prologue's fma / mul as scalar VALU instead of v_pk_*): isolated + e2e A/B
O=gpurun_out/${TAG:-r5_noslp}; mkdir -p $O
C=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
export MI355_CONV_TIME=100
{
for rep in 1 2; do
  for shape in "256 256 16 256 3" "256 512 16 256 3" "256 128 32 128 3" "256 256 16 256 3 nogn"; do
    echo -n "shape $shape base: "; MI355_CONV_PP=29 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1
    echo -n "shape $shape noslp: "; MI355_SAMPLER_LIB=$C/libmi355_sampler_a0_p0_noslp.so MI355_CONV_PP=29 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1
  done
done
} 2>&1 | tee $O/times.txt
unset MI355_CONV_TIME
for rep in 1 2 3; do
  echo -n "base: "; python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'img/s')"
  echo -n "noslp: "; MI355_SAMPLER_LIB=$C/libmi355_sampler_a0_p0_noslp.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'img/s')"
done 2>&1 | tee $O/bench_ab.txt

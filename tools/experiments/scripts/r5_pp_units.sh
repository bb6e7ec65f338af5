#!/bin/bash
# round 5: half-piece units vs whole pieces vs no prologue math vs the round-4 path (ws with its loader prologue), isolated same-box timings, interleaved x2
O=gpurun_out/${TAG:-r5_pp_units}; mkdir -p $O
C=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "pingpong" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -3 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
export MI355_CONV_TIME=100
{
for rep in 1 2; do
  for shape in "256 128 32 128 3" "256 256 32 128 3" "256 384 32 128 3" "256 128 16 256 3" "256 256 16 256 3" "256 512 16 256 3"; do
    echo -n "shape $shape ws: "; MI355_CONV_PP=1 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
    echo -n "shape $shape units: "; MI355_CONV_PP=13 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
    echo -n "shape $shape whole: "; MI355_SAMPLER_LIB=$C/libmi355_sampler_a0_p0_whole.so MI355_CONV_PP=13 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
    echo -n "shape $shape nomath: "; MI355_SAMPLER_LIB=$C/libmi355_sampler_a0_p256_nomath.so MI355_CONV_PP=13 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1 || echo fail
  done
done
} 2>&1 | tee $O/times.txt

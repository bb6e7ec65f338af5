#!/bin/bash
# round 4: one GPU round for the ping-pong conv: parity tests, isolated launch times ws / pp, stamps + timeline
O=gpurun_out/${TAG:-r4_pp2}; mkdir -p $O
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -k "pingpong" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -4 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
run() { echo -n "$SHAPE | $1: "; shift; env "$@" MI355_CONV_TIME=50 timeout -k 10 120 python tools/time_conv.py $SHAPE nogn 2>&1 | grep -E "conv time" | tail -1 | sed 's/\[conv time\] 50 launches, //'; }
IFS=";" read -ra SH <<< "${SHAPES:-256 256 16 256 3;256 512 16 256 3;256 128 16 256 3;512 256 16 256 3}"; unset IFS
{
for SHAPE in "${SH[@]}"; do
  for rep in 1 2; do
    run "ws " MI355_CONV_PP=0
    run "pp " MI355_CONV_PP=2
  done
done
} 2>&1 | tee $O/times.txt
SHAPES="256 256 16 256 3;256 512 16 256 3" REPS=50 TAG=${TAG:-r4_pp2} bash tools/r4_pp_stamps.sh

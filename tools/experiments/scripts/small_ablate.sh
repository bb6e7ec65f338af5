#!/bin/bash
# Diagnostic: launch time of the small-level conv (conv_small.inc.h) under baked-in ablation masks (csrc: make variant ABL=<mask>:
# 4 = no LDS reads / MFMAs, 8 = no weight loads, 16 = no patch staging) and MI355_CONV_ABLATE=1 (no output stores).
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
run() { echo -n "$1: "; shift; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE nogn 2>&1 | grep "conv time" | tail -1; }
for SHAPE in "256 256 8 256 3" "256 512 8 256 3" "256 256 4 256 3" "256 512 4 256 3" "1024 256 8 256 3"; do
  echo "== $SHAPE"
  run "plain kernel    " MI355_CONV_SMALL=0
  run "small           " MI355_CONV_SMALL=1
  run "small no stores " MI355_CONV_ABLATE=1
  for v in "$@"; do run "small variant $v" MI355_SAMPLER_LIB=$D/libmi355_sampler_$v.so; done
done

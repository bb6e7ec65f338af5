#!/bin/bash
# Diagnostic: launch time of the warp-specialised conv under baked-in ablation masks (csrc: make variant ABL=<mask>)
# and under the runtime masks (MI355_CONV_ABLATE: 1 = no output stores, 2 = no GN/SiLU prologue math).
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
SHAPE=${SHAPE:-"256 128 32 128 3"}
run() { echo -n "$1: "; shift; env "$@" MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep "conv time" | tail -1; }
run "plain kernel   " MI355_CONV_WS=0
run "ws             " MI355_CONV_WS=1
run "ws no stores   " MI355_CONV_ABLATE=1
run "ws no prologue " MI355_CONV_ABLATE=2
run "ws no st/pro   " MI355_CONV_ABLATE=3
for v in "$@"; do run "ws variant $v" MI355_SAMPLER_LIB=$D/libmi355_sampler_$v.so; run "ws variant $v no st/pro" MI355_SAMPLER_LIB=$D/libmi355_sampler_$v.so MI355_CONV_ABLATE=3; done

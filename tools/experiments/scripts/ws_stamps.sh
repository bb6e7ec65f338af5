#!/bin/bash
# Diagnostic: in-kernel phase stamps of the warp-specialised conv (csrc: make stamps)
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
SHAPE=${SHAPE:-"256 128 32 128 3"}
MI355_SAMPLER_LIB=$D/libmi355_sampler_stamps${V}.so timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep "conv stamps" | tail -1

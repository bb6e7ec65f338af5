#!/bin/bash
# Diagnostic: wave-priority settings of the warp-specialised conv (MI355_CONV_STAGGER = consumer prio | loader prio << 2; 16 = both 0)
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
for SHAPE in "256 128 32 128 3" "256 256 16 256 3"; do
for lib in libmi355_sampler.so libmi355_sampler_old.so; do
echo "== $SHAPE $lib"
for st in 2 16 1 3 4 8 12 6 9 14 13; do echo -n "consumer prio $((st & 3)) loader prio $(((st >> 2) & 3)): "; MI355_SAMPLER_LIB=$D/$lib MI355_CONV_STAGGER=$st MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep "conv time" | tail -1; done
done
done

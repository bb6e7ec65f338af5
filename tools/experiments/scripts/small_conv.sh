#!/bin/bash
# Diagnostic: tile choice for the small-image 3x3 convs (MI355_CONV_MINWG = workgroups a launch must have before a tile is accepted)
for shape in "256 256 4 256 3" "256 512 4 256 3" "256 256 8 256 3" "256 512 8 256 3"; do
  for mw in 256 512 1024 2048; do echo -n "$shape minwg=$mw: "; MI355_CONV_MINWG=$mw MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $shape 2>&1 | grep "conv time" | tail -1; done
done

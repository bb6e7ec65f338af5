#!/bin/bash
# in-kernel phase stamps (diagnostic build) of the small-level 3x3 convs at the bench batch
S=$GRAFT_REPO_ROOT/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc/libmi355_sampler_stamps.so
for shape in "256 256 8 256 3" "256 512 8 256 3" "256 256 4 256 3"; do
  echo "== $shape (stamps)"; MI355_SAMPLER_LIB=$S MI355_CONV_TIME=20 timeout -k 10 120 python tools/time_conv.py $shape nogn 2>&1 | grep -E "conv stamps|conv time"
  echo "== $shape (timed build)"; MI355_CONV_TIME=50 timeout -k 10 120 python tools/time_conv.py $shape nogn 2>&1 | grep -E "conv time"
done

#!/bin/bash
# in-kernel clock of the persistent conv's consumer waves after >= 2 s of back-to-back launches (stamps build: make stamps), with the
# launch time of the same run; and the same for the timed library's launch time alone
D=$PWD/image-inpainting-and-super-resolution-using-diffusion-models-and-conditional-flow-matching_amd/csrc
O=gpurun_out/r3_clock; mkdir -p $O
for SHAPE in "256 128 32 128 3" "256 256 16 256 3" "256 512 16 256 3"; do for GNV in "" nogn; do
  echo "== $SHAPE $GNV"
  MI355_SAMPLER_LIB=$D/libmi355_sampler_stamps.so MI355_CONV_TIME=20000 timeout -k 10 200 python tools/time_conv.py $SHAPE $GNV 2>&1 | grep -E "conv clock|conv time" | sed 's/^/stamps build: /'
  MI355_CONV_TIME=20000 timeout -k 10 200 python tools/time_conv.py $SHAPE $GNV 2>&1 | grep -E "conv time" | sed 's/^/timed build:  /'
done; done | tee $O/clock.txt

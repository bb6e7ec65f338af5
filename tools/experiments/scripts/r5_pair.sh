#!/bin/bash
# round 5 (last): pair form of the 8x8-level conv (conv_small bit 4): isolated launches and per-op events in the network, one box
O=gpurun_out/${TAG:-r5b_pair2}; mkdir -p $O
{
for shape in "256 256 8 256 3" "256 512 8 256 3"; do
  for cs in 13 15 31; do
    echo -n "isolated $shape conv_small=$cs: "
    MI355_CONV_TIME=100 MI355_CONV_SMALL=$cs python tools/time_conv.py $shape nogn 2>&1 | grep "conv time" | tail -1
  done
done
} 2>&1 | tee $O/isolated.txt
for cs in 15 31; do
  MI355_CONV_SMALL=$cs python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-out $O/per_op_$cs.json > /dev/null 2>&1
  python tools/show_profile.py $O/per_op_$cs.json | grep -E "forward|8x8" | tee $O/per_op_$cs.txt
done

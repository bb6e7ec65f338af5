#!/bin/bash
# round 4: runtime ablations of the ping-pong conv (timed build; results of ablated runs are wrong by construction)
O=gpurun_out/${TAG:-r4_pp_ablate}; mkdir -p $O
IFS=";" read -ra SH <<< "${SHAPES:-256 256 16 256 3;256 512 16 256 3}"; unset IFS
for SHAPE in "${SH[@]}"; do
for A in ${ABLS:-0 1 4 8 16 24 128 132 28 156 0}; do
echo -n "$SHAPE ablate=$A: "; MI355_CONV_PP=2 MI355_CONV_ABLATE=$A MI355_CONV_TIME=${REPS:-100} timeout -k 10 120 python tools/time_conv.py $SHAPE nogn 2>&1 | grep -E "conv time" | sed 's/\[conv time\] //'
done; done 2>&1 | tee $O/ablate.txt

#!/bin/bash
# Diagnostic: wave-priority settings of the warp-specialised conv (MI355_CONV_STAGGER = consumer prio | loader prio << 2; 16 = both 0)
SHAPE=${SHAPE:-"256 128 32 128 3"}
for st in 2 16 1 3 6 9 14 13; do echo -n "stagger=$st: "; MI355_CONV_STAGGER=$st MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE 2>&1 | grep "conv time" | tail -1; done

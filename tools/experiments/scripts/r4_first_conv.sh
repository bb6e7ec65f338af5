#!/bin/bash
# round 4: first-conv kernel: parity tests, then isolated and end-to-end A/B on one box
O=gpurun_out/${TAG:-r4_first_conv}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu -k "first_conv" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -3 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
TAG=${TAG:-r4_first_conv} TESTS="" ENVS="old:MI355_CONV_EDGE=1;new:MI355_CONV_EDGE=3" PROFILE=1 bash tools/r4_e2e_ab.sh

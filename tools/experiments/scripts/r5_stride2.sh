#!/bin/bash
# round 5: stride-2 Downsample convs 16 -> 8 / 8 -> 4 on the small-level kernel (conv_small bit 2): parity, isolated timings, e2e
O=gpurun_out/${TAG:-r5_stride2}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_unet.py -x -q -m gpu -k "stride2 or golden or cfg2_b256" > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt; tail -3 $O/test.txt
grep -q "rc=0" $O/test.txt || exit 1
for rep in 1 2 3; do for sm in 3 7; do echo -n "small=$sm: "; MI355_CONV_SMALL=$sm python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'img/s')"; done; done 2>&1 | tee $O/bench_ab.txt
for sm in 3 7; do MI355_CONV_SMALL=$sm python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-out $O/per_op_$sm.json > /dev/null 2>&1; python tools/show_profile.py $O/per_op_$sm.json | grep -E "8x8|4x4" | grep "k3" > $O/per_op_$sm.txt; done
paste -d'\n' $O/per_op_3.txt /dev/null $O/per_op_7.txt | head -30

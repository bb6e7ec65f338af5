#!/bin/bash
# attention_kernel after the register-budget / mask / lazy-rescale change: parity, then kernel durations in the 128-px workload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_attn; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -x -q -m gpu -k "attention or attn or cfg5 or cfg3 or px128" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
grep -q " passed" $O/tests.txt || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/px128 -o s -- python3 $R/bench.py --workload px128_inpaint_ddim100_b128 --steps 1 --warmup 1 --nfe 2 --no-cpu-baseline > $O/px128.log 2>&1 || exit 1
cd $R
python tools/rocpd_stats.py $O/px128/s_results.db $O/px128.csv; rm -rf $O/px128
grep -i "attention_kernel\|attn_fused" $O/px128.csv | cut -c1-160
python bench.py --workload px128_inpaint_ddim100_b128 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-150

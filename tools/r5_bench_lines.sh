#!/bin/bash
# Bench lines of every workload / precision on one box (the first part of tools/r5_final_profiles.sh, without the rocprofv3 passes)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r5b_lines}
mkdir -p $O
cd $R
rm -f $O/bench_lines_all_workloads.jsonl $O/bench_log.txt
run() { echo "== $*" >> $O/bench_log.txt; python bench.py "$@" 2>>$O/bench_log.txt | tee -a $O/bench_lines_all_workloads.jsonl | cut -c1-120; }
run --steps 10 --warmup 3 || exit 1
run --steps 2 --warmup 1 --precision fp32 --no-cpu-baseline || exit 1
run --steps 5 --warmup 2 --precision bf16x2 --no-cpu-baseline || exit 1
run --steps 5 --warmup 2 --precision fp16 --no-cpu-baseline || exit 1
run --steps 3 --warmup 1 --batch 512 --no-cpu-baseline || exit 1
run --steps 3 --warmup 1 --workload cifar10_inpaint_ddpm50_b512 --no-cpu-baseline || exit 1
run --steps 3 --warmup 1 --workload cifar10_inpaint_ddim50_b512 --no-cpu-baseline || exit 1
run --steps 2 --warmup 1 --workload cifar64_cfm_euler50_b256 --no-cpu-baseline || exit 1
run --steps 2 --warmup 1 --workload flowers64_superres_euler100_b256 --no-cpu-baseline || exit 1
run --steps 2 --warmup 1 --workload px128_inpaint_ddim100_b128 --no-cpu-baseline || exit 1
echo "bench lines done"

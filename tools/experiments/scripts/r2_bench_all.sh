#!/bin/bash
# Round-2 measurement sweep: every bench workload (one JSON line each) + fp32 mode + larger batches of the headline config.
set -o pipefail
out=gpurun_out/r2_bench
mkdir -p $out
run() { echo "== $*" | tee -a $out/log.txt; python bench.py "$@" 2>>$out/log.txt | tee -a $out/lines.jsonl | cut -c1-400; }
run --steps 5 --warmup 2 --profile-out $out/per_op_cifar_b256.json || exit 1
run --steps 2 --warmup 1 --precision fp32 --no-cpu-baseline || exit 1
run --steps 3 --warmup 1 --batch 512 --no-cpu-baseline || exit 1
run --steps 3 --warmup 1 --batch 1024 --no-cpu-baseline || exit 1
run --steps 3 --warmup 1 --workload cifar10_inpaint_ddpm50_b512 || exit 1
run --steps 3 --warmup 1 --workload cifar10_inpaint_ddim50_b512 || exit 1
run --steps 2 --warmup 1 --workload cifar64_cfm_euler50_b256 --profile-out $out/per_op_cifar64_b256.json || exit 1
run --steps 2 --warmup 1 --workload flowers64_superres_euler100_b256 --profile-out $out/per_op_flowers64_b256.json || exit 1
run --steps 2 --warmup 1 --workload px128_inpaint_ddim100_b128 --profile-out $out/per_op_px128_b128.json || exit 1

#!/bin/bash
# Round-5 final evidence on the final build: bench lines of every workload (+ per-op profiles), rocprofv3 kernel stats of the headline
# command and of the 64x64 / 128-px workloads, PMC passes (SQ counters; FETCH_SIZE and WRITE_SIZE in separate passes).
# Summaries land in gpurun_out/r5_final/ (copied into profiles/ afterwards).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_final
mkdir -p $O
cd $R
rm -f $O/bench_lines_all_workloads.jsonl $O/bench_log.txt
run() { echo "== $*" >> $O/bench_log.txt; python bench.py "$@" 2>>$O/bench_log.txt | tee -a $O/bench_lines_all_workloads.jsonl | cut -c1-160; }
run --steps 10 --warmup 3 --profile-out $O/per_op_cifar_b256.json || exit 1
run --steps 2 --warmup 1 --precision fp32 --no-cpu-baseline || exit 1
run --steps 5 --warmup 2 --precision bf16x2 --no-cpu-baseline || exit 1
run --steps 5 --warmup 2 --precision fp16 --no-cpu-baseline || exit 1
run --steps 3 --warmup 1 --batch 512 --no-cpu-baseline || exit 1
run --steps 3 --warmup 1 --workload cifar10_inpaint_ddpm50_b512 || exit 1
run --steps 3 --warmup 1 --workload cifar10_inpaint_ddim50_b512 || exit 1
run --steps 2 --warmup 1 --workload cifar64_cfm_euler50_b256 --profile-out $O/per_op_cifar64_b256.json || exit 1
run --steps 2 --warmup 1 --workload flowers64_superres_euler100_b256 --profile-out $O/per_op_flowers64_b256.json || exit 1
run --steps 2 --warmup 1 --workload px128_inpaint_ddim100_b128 --profile-out $O/per_op_px128_b128.json || exit 1
echo "bench lines done" 
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 1 --nfe 4 --no-cpu-baseline"
# kernel stats: the bench command at its full length (100 network evaluations) - a 8-evaluation run is over before the clocks have
# settled, and its averages read ~9 % above the live HIP-event figure of the bench line
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/sq1 -o s -- $CMD > $O/sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES -d $O/sq2 -o s -- $CMD > $O/sq2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1 || exit 1
echo "pmc passes done"
rocprofv3 --kernel-trace --stats -d $O/px128 -o s -- python3 $R/bench.py --workload px128_inpaint_ddim100_b128 --steps 1 --warmup 1 --nfe 2 --no-cpu-baseline > $O/px128.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $O/fl64 -o s -- python3 $R/bench.py --workload flowers64_superres_euler100_b256 --steps 1 --warmup 1 --nfe 2 --no-cpu-baseline > $O/fl64.log 2>&1 || exit 1
cd $R
python tools/rocpd_stats.py $O/stats/s_results.db $O/rocprofv3_kernel_stats_bench_cifar_b256.csv
python tools/pmc_kernels.py $O/sq1/s_results.db $O/pmc_sq_wave_valu_wait.json
python tools/pmc_kernels.py $O/sq2/s_results.db $O/pmc_sq_mfma_lds.json
python tools/pmc_traffic.py $O/fetch/f_results.db $O/write/w_results.db $O/pmc_hbm_traffic.json
python tools/rocpd_stats.py $O/px128/s_results.db $O/rocprofv3_kernel_stats_px128_b128.csv
python tools/rocpd_stats.py $O/fl64/s_results.db $O/rocprofv3_kernel_stats_flowers64_b256.csv
rm -rf $O/stats $O/sq1 $O/sq2 $O/fetch $O/write $O/px128 $O/fl64
ls $O

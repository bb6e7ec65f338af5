#!/bin/bash
# Round-2 final evidence: rocprofv3 kernel stats + PMC (SQ, HBM traffic) of the headline bench command, and kernel stats of the
# cfg 5 geometry (128 px, attention at T = 1024 / 256 / 64).  Summaries land in gpurun_out/r2_final/ (copied into profiles/).
set -o pipefail
bash tools/r2_pmc.sh || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_final
mkdir -p $O
cp $R/gpurun_out/r2_prof/kernel_stats.csv $O/rocprofv3_kernel_stats_bench_cifar_b256.csv
cp $R/gpurun_out/r2_prof/sq1.json $O/pmc_sq_wave_valu_wait.json
cp $R/gpurun_out/r2_prof/sq2.json $O/pmc_sq_mfma_lds.json
cp $R/gpurun_out/r2_prof/hbm_traffic.json $O/pmc_hbm_traffic.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/px128 -o s -- python3 $R/bench.py --workload px128_inpaint_ddim100_b128 --steps 1 --warmup 1 --nfe 2 --no-cpu-baseline > $O/px128.log 2>&1 || exit 1
cd $R && python tools/rocpd_stats.py $O/px128/s_results.db $O/rocprofv3_kernel_stats_px128_b128.csv && rm -rf $O/px128

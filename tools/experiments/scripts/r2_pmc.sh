#!/bin/bash
# rocprofv3 passes of the headline bench command (short: 1 step of 4 network evaluations): kernel stats, SQ counters, HBM traffic.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_prof
mkdir -p $O
CMD="python3 $R/bench.py --steps 1 --warmup 1 --nfe 4 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $CMD > $O/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $O/sq1 -o s -- $CMD > $O/sq1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES -d $O/sq2 -o s -- $CMD > $O/sq2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1 || exit 1
cd $R
python tools/rocpd_stats.py $O/stats/s_results.db $O/kernel_stats.csv
python tools/pmc_kernels.py $O/sq1/s_results.db $O/sq1.json
python tools/pmc_kernels.py $O/sq2/s_results.db $O/sq2.json
python tools/pmc_traffic.py $O/fetch/f_results.db $O/write/w_results.db $O/hbm_traffic.json
rm -rf $O/stats $O/sq1 $O/sq2 $O/fetch $O/write

#!/bin/bash
# VERDICT r2 task 8 evidence: kernel summary of one super-resolution solve (40 network evaluations): at::native kernels appear a
# handful of times (seeding x0, the final clip), never once per evaluation
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_superres; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/t -o s -- python3 $R/tools/superres_solve.py --steps 40 > $O/solve.log 2>&1 || { tail -5 $O/solve.log; exit 1; }
cd $R
grep "solve done" $O/solve.log
python tools/rocpd_stats.py $O/t/s_results.db $O/kernel_stats_utils_mnist_hy2_generate_samples_eval_euler40.csv
rm -rf $O/t
cut -d, -f1,2 $O/kernel_stats_utils_mnist_hy2_generate_samples_eval_euler40.csv | cut -c1-120

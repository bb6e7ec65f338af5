#!/bin/bash
# per-kernel durations of three variants: finalize by launch / in-kernel / none (wrong results)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_fin_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 1 --nfe 10 --no-cpu-baseline"
export MI355_GN_FIN_INLINE=0
rocprofv3 --kernel-trace --stats -d $O/launch -o s -- $CMD > $O/launch.log 2>&1 || exit 1
export MI355_GN_FIN_INLINE=1
rocprofv3 --kernel-trace --stats -d $O/inline -o s -- $CMD > $O/inline.log 2>&1 || exit 1
export MI355_CONV_ABLATE=4096
rocprofv3 --kernel-trace --stats -d $O/nofin -o s -- $CMD > $O/nofin.log 2>&1 || exit 1
cd $R
for v in launch inline nofin; do python tools/rocpd_stats.py $O/$v/s_results.db $O/$v.csv; python3 - $O/$v/s_results.db <<'PY'
import sqlite3,sys
cur=sqlite3.connect(sys.argv[1]).cursor()
rows=sorted(cur.execute("select start,end,name from kernels").fetchall())
# the last forward: busy time and gaps over the last 134*5 dispatches
rows=rows[-600:]
busy=sum(e-s for s,e,_ in rows); span=rows[-1][1]-rows[0][0]
print("last 600 dispatches: span %.1f us busy %.1f us gaps %.1f us" % (span/1e3,busy/1e3,(span-busy)/1e3))
PY
rm -rf $O/$v; done

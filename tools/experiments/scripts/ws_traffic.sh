#!/bin/bash
# HBM traffic per launch of the persistent conv on isolated shapes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around tools/time_conv.py
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ws_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SHAPE in "256 256 16 256 3" "256 128 32 128 3" "256 512 16 256 3"; do
  i=$((i+1))
  MI355_CONV_TIME=10 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f$i -o f -- python3 $R/tools/time_conv.py $SHAPE > $O/f$i.log 2>&1 || exit 1
  MI355_CONV_TIME=10 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w$i -o w -- python3 $R/tools/time_conv.py $SHAPE > $O/w$i.log 2>&1 || exit 1
  (cd $R && python tools/pmc_traffic.py $O/f$i/f_results.db $O/w$i/w_results.db $O/t$i.json > /dev/null && python - <<P
import json
d=json.load(open("$O/t$i.json"))["kernels"]
for k,v in d.items():
    if "ws_kernel" in k: print("$SHAPE", k, {a:round(b/1e6,1) if isinstance(b,float) else b for a,b in v.items()})
P
)
  rm -rf $O/f$i $O/w$i
done

#!/bin/bash
# round 3, experiment 1: what a once-per-tensor GN-apply pass + prologue-free convs cost with the code as it stands
# (MI355_GN_APPLY_MAXHW lifts the small-image apply path to every GroupNorm site; the convs then run PRO = 0, register-staged)
set -o pipefail
O=gpurun_out/r3_exp1; mkdir -p $O
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-out $O/prof_base.json > $O/bench_base.json 2> $O/bench_base.err || exit 1
MI355_GN_APPLY_MAXHW=4096 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-out $O/prof_apply.json > $O/bench_apply.json 2> $O/bench_apply.err || exit 1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_base2.json 2>> $O/bench_base.err || exit 1
for SHAPE in "256 128 32 128 3" "256 256 16 256 3" "256 512 16 256 3" "256 256 32 128 3" "256 384 32 128 3"; do
  for V in "" "nogn"; do
    echo -n "$SHAPE $V: " >> $O/conv_times.txt
    MI355_CONV_TIME=30 timeout -k 10 120 python tools/time_conv.py $SHAPE $V 2>&1 | grep "conv time" | tail -1 >> $O/conv_times.txt
  done
done
cut -c1-200 $O/bench_base.json $O/bench_apply.json $O/bench_base2.json
cat $O/conv_times.txt

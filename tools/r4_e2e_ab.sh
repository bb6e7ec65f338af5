#!/bin/bash
# round 4: GPU test suites, then end-to-end bench A/B over environment settings on one box (interleaved)
#   ENVS="name:VAR=val,VAR2=val;name2:..."   TESTS="<pytest files>" (empty = skip)  KEXPR="<pytest -k expression>"
O=gpurun_out/${TAG:-r4_e2e}; mkdir -p $O
if [ -n "$TESTS" ]; then
  if [ -n "$KEXPR" ]; then timeout -k 10 1000 python -m pytest $TESTS -x -q -rP -m gpu -k "$KEXPR" > $O/test.txt 2>&1; else timeout -k 10 1000 python -m pytest $TESTS -x -q -m gpu > $O/test.txt 2>&1; fi; echo "pytest rc=$?" >> $O/test.txt; tail -4 $O/test.txt
  grep -q "rc=0" $O/test.txt || exit 1
fi
IFS=";" read -ra EV <<< "${ENVS:-pp0:MI355_CONV_PP=0;pp1:MI355_CONV_PP=1}"; unset IFS
{
for rep in 1 2 3; do
  for E in "${EV[@]}"; do
    name=${E%%:*}; vars=${E#*:}
    echo -n "$name: "; env ${vars//,/ } python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline ${BENCHARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
  done
done
} 2>&1 | tee $O/bench_ab.txt
if [ -n "$PROFILE" ]; then python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-out $O/per_op.json > /dev/null 2>&1; python tools/show_profile.py $O/per_op.json > $O/per_op.txt; head -45 $O/per_op.txt; fi

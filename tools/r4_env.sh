#!/bin/bash
out=gpurun_out/r4s
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'])"; }
for i in 1 2; do
echo "wrn-50-2b default" >> $out/env.log; run --workload wrn-50-2b --steps 10 >> $out/env.log
echo "wrn-50-2b RN_VARIANT=536870912 (no wgrad8: round-2 weight gradient, side-friendly BN)" >> $out/env.log; RN_VARIANT=536870912 run --workload wrn-50-2b --steps 10 >> $out/env.log
echo "wrn-50-2b RN_BN_LIGHT=0" >> $out/env.log; RN_BN_LIGHT=0 run --workload wrn-50-2b --steps 10 >> $out/env.log
done
cat $out/env.log

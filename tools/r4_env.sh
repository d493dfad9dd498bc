#!/bin/bash
out=gpurun_out/r4i
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'])"; }
for i in 1 2; do
echo "default" >> $out/env.log; run >> $out/env.log
echo "RN_NO_DGRAD_FUSION=1" >> $out/env.log; RN_NO_DGRAD_FUSION=1 run >> $out/env.log
done
echo "RN_NO_OVERLAP=1" >> $out/env.log; RN_NO_OVERLAP=1 run >> $out/env.log
echo "RN_NO_OVERLAP=1 RN_NO_DGRAD_FUSION=1" >> $out/env.log; RN_NO_OVERLAP=1 RN_NO_DGRAD_FUSION=1 run >> $out/env.log
cat $out/env.log

#!/bin/bash
out=gpurun_out/r4n
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'])"; }
for i in 1 2; do
for g in 256 248 240 232; do
echo "RN_W8_FORK_GRID=$g" >> $out/env.log; RN_W8_FORK_GRID=$g run >> $out/env.log
done
echo "RN_VARIANT2=16388" >> $out/env.log; RN_VARIANT2=16388 run >> $out/env.log
done
cat $out/env.log

#!/bin/bash
out=gpurun_out/r4z
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'], d['roofline']['frac'])"; }
for i in 1 2 3; do
echo "shipped" >> $out/env.log; run >> $out/env.log
echo "RN_NO_S2_DGRAD_FUSION=1" >> $out/env.log; RN_NO_S2_DGRAD_FUSION=1 run >> $out/env.log
done
cat $out/env.log
RN_NO_S2_DGRAD_FUSION=1 python bench.py --no-cpu-baseline --no-parity --also= --breakdown --per-op 100 2>&1 >/dev/null | grep -E "s2|BN_BWD_REDUCE" | cut -c1-130 | head -8

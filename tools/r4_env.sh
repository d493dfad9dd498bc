#!/bin/bash
# batch size of the forked weight-gradient launches / their grid cap: bench lines by environment, one box
out=gpurun_out/r4g
mkdir -p $out
for b in 4 2 8 12; do
  echo "RN_W8R_BATCH=$b" >> $out/env.log
  RN_W8R_BATCH=$b timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity --also= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread'])" >> $out/env.log
done
for g in 224 192; do
  echo "RN_W8_FORK_GRID=$g" >> $out/env.log
  RN_W8_FORK_GRID=$g timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity --also= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread'])" >> $out/env.log
done
echo "RN_W8R_BATCH=4 again" >> $out/env.log
RN_W8R_BATCH=4 timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity --also= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread'])" >> $out/env.log
cat $out/env.log

#!/bin/bash
out=gpurun_out/r4h
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_production_tiles.py -x -q -m gpu -k "igemm8r or wgrad9 or wgrad8r" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
RN_CONV_VARIANT="0/4098,0/2" timeout -k 10 200 python tools/conv_bench.py fwd 20 >> $out/ab.log 2>&1
grep -v amdgpu.ids $out/ab.log
for b in 8 4 12 8; do
  echo "RN_W8R_BATCH=$b" >> $out/env.log
  RN_W8R_BATCH=$b timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity --also= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread'])" >> $out/env.log
done
cat $out/env.log

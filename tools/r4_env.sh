#!/bin/bash
out=gpurun_out/r4q
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'])"; }
for i in 1 2; do
echo "default (side-friendly BN, batch 1)" >> $out/env.log; run >> $out/env.log
echo "RN_BN_LIGHT=0 RN_W8R_BATCH=8 (before)" >> $out/env.log; RN_BN_LIGHT=0 RN_W8R_BATCH=8 run >> $out/env.log
echo "RN_VARIANT2=16389 RN_BN_LIGHT=0 (round-3 kernels)" >> $out/env.log; RN_VARIANT2=16389 RN_BN_LIGHT=0 run >> $out/env.log
echo "RN_VARIANT2=16389 (round-3 kernels, side-friendly BN)" >> $out/env.log; RN_VARIANT2=16389 run >> $out/env.log
done
for wl in v2-164 rn20; do
echo "$wl default" >> $out/env.log; run --workload $wl >> $out/env.log
echo "$wl RN_BN_LIGHT=0" >> $out/env.log; RN_BN_LIGHT=0 run --workload $wl >> $out/env.log
done
echo "wrn-50-2b default" >> $out/env.log; run --workload wrn-50-2b --steps 10 >> $out/env.log
echo "wrn-50-2b RN_BN_LIGHT=0" >> $out/env.log; RN_BN_LIGHT=0 run --workload wrn-50-2b --steps 10 >> $out/env.log
cat $out/env.log
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log

#!/bin/bash
out=gpurun_out/r4r
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'])"; }
for i in 1 2 3; do
echo "default (slab sums on side2)" >> $out/env.log; run >> $out/env.log
echo "RN_NO_SIDE2=1" >> $out/env.log; RN_NO_SIDE2=1 run >> $out/env.log
echo "RN_NO_DGRAD_FUSION=1" >> $out/env.log; RN_NO_DGRAD_FUSION=1 run >> $out/env.log
done
cat $out/env.log
timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_ddp_gloo.py -x -q -m gpu > $out/tests.log 2>&1; tail -3 $out/tests.log

#!/bin/bash
out=gpurun_out/r4w
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_production_tiles.py -x -q -m gpu -k "wgrad9" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
RN_CONV_VARIANT="0/8,0/32776" timeout -k 10 200 python tools/conv_bench.py wgrad 20 > $out/ab.log 2>&1; grep -v amdgpu.ids $out/ab.log
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'])"; }
for i in 1 2 3; do
echo "shipped" >> $out/env.log; run >> $out/env.log
echo "RN_VARIANT2=32768 (wgrad9: two K tiles per barrier)" >> $out/env.log; RN_VARIANT2=32768 run >> $out/env.log
echo "RN_DROP_RECOMPUTE=1" >> $out/env.log; RN_DROP_RECOMPUTE=1 run >> $out/env.log
done
cat $out/env.log
RN_DROP_RECOMPUTE=1 timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "dropout or sgd or learns" 2>&1 | tail -2

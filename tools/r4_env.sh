#!/bin/bash
out=gpurun_out/r4y
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_production_tiles.py -x -q -m gpu -k "wgrad9 or wgrad8r" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'], d['roofline']['frac'])"; }
for i in 1 2 3; do
echo "shipped (stride-2 weight gradients on wgrad9)" >> $out/env.log; run >> $out/env.log
echo "RN_VARIANT2=524288 (on wgrad8r)" >> $out/env.log; RN_VARIANT2=524288 run >> $out/env.log
done
cat $out/env.log
python bench.py --no-cpu-baseline --no-parity --also= --breakdown --per-op 100 2>&1 >/dev/null | grep -E "s2" | cut -c1-130
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "wrn28" 2>&1 | tail -2

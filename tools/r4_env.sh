#!/bin/bash
out=gpurun_out/r4aa
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_production_tiles.py -x -q -m gpu -k "stride2 or epilogue or production_tiles or classes or proj or dgrad" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'], d['roofline']['frac'])"; }
for i in 1 2; do
echo "wrn-50-2b shipped (tap-less classes as a pass)" >> $out/env.log; run --workload wrn-50-2b --steps 10 >> $out/env.log
echo "wrn-50-2b RN_VARIANT2=1048576 (in the convolution kernels)" >> $out/env.log; RN_VARIANT2=1048576 run --workload wrn-50-2b --steps 10 >> $out/env.log
done
for i in 1 2; do
echo "wrn-28-10 shipped" >> $out/env.log; run >> $out/env.log
echo "wrn-28-10 RN_VARIANT2=1048576" >> $out/env.log; RN_VARIANT2=1048576 run >> $out/env.log
done
cat $out/env.log
python bench.py --workload wrn-50-2b --steps 5 --no-cpu-baseline --no-parity --also= --breakdown --per-op 60 2>&1 >/dev/null | grep -E "k1 s2" | cut -c1-140

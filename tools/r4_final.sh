#!/bin/bash
# round-4 record on ONE box: shipped configuration against (a) the wide BatchNorm kernels + 8-layer batches (the state before R4-m), (b) round 3's kernel selection
out=gpurun_out/r4t
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'])"; }
for i in 1 2 3; do
echo "shipped" >> $out/ab.log; run >> $out/ab.log
echo "RN_BN_LIGHT=0 RN_W8R_BATCH=8 RN_NO_SIDE2=1 (before R4-m)" >> $out/ab.log; RN_BN_LIGHT=0 RN_W8R_BATCH=8 RN_NO_SIDE2=1 run >> $out/ab.log
echo "RN_VARIANT2=16389 RN_BN_LIGHT=0 (round-3 kernels)" >> $out/ab.log; RN_VARIANT2=16389 RN_BN_LIGHT=0 run >> $out/ab.log
done
echo "RN_NO_OVERLAP=1" >> $out/ab.log; RN_NO_OVERLAP=1 run >> $out/ab.log
for wl in wrn-50-2b v2-164 rn20; do echo "$wl" >> $out/ab.log; run --workload $wl --steps 10 >> $out/ab.log; done
cat $out/ab.log
timeout -k 10 500 python bench.py --breakdown --per-op 100 > $out/bench.json 2> $out/per_op.txt
cat $out/bench.json

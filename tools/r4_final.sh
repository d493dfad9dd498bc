#!/bin/bash
# round-4 record on ONE box: the shipped kernel selection against round 3's (RN_VARIANT2 = 1 | 4 | 16384: no igemm8r / wgrad8r / wgrad9), alternating; the
# timing probes of the shipped weight-gradient schedule; the per-op listing
out=gpurun_out/r4k
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'])"; }
for i in 1 2 3; do
echo "round-4 kernels" >> $out/ab.log; run >> $out/ab.log
echo "round-3 kernels (RN_VARIANT2=16389)" >> $out/ab.log; RN_VARIANT2=16389 run >> $out/ab.log
done
cat $out/ab.log
RN_CONV_VARIANT="0/8,0/264,0/520,0/776" timeout -k 10 200 python tools/conv_bench.py wgrad 20 > $out/probes.log 2>&1
RN_CONV_VARIANT="0/2,0/2097154" timeout -k 10 200 python tools/conv_bench.py fwd 20 >> $out/probes.log 2>&1
grep -v amdgpu.ids $out/probes.log
timeout -k 10 500 python bench.py --breakdown --per-op 100 > $out/bench.json 2> $out/per_op.txt
cut -c1-700 $out/bench.json

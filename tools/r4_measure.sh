#!/bin/bash
# round-4 state check: micro-benchmarks old vs new, the default bench line with the per-op listing
set -e
out=gpurun_out/r4a
mkdir -p $out
for ep in 0 1; do
echo "EP=$ep" >> $out/conv_bench.log
RN_CONV_EP=$ep RN_CONV_VARIANT="0/1,0/2" timeout -k 10 200 python tools/conv_bench.py fwd 20 >> $out/conv_bench.log 2>&1
RN_CONV_EP=$ep RN_CONV_VARIANT="0/1,0/2" timeout -k 10 200 python tools/conv_bench.py dgrad 20 >> $out/conv_bench.log 2>&1
done
RN_CONV_VARIANT="0/16396,0/16392,0/8" timeout -k 10 200 python tools/conv_bench.py wgrad 20 >> $out/conv_bench.log 2>&1
cat $out/conv_bench.log
timeout -k 10 500 python bench.py --breakdown --per-op 100 > $out/bench.json 2> $out/bench.err
cat $out/bench.json

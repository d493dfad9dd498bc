#!/bin/bash
out=gpurun_out/r4e
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_production_tiles.py -x -q -m gpu -k "igemm8r" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for ep in 0 1; do
echo "EP=$ep" >> $out/ab.log
RN_CONV_SHAPES="128,8,8,640,640,3" RN_CONV_EP=$ep RN_CONV_VARIANT="0/131072,0/0,0/131074" timeout -k 10 200 python tools/conv_bench.py fwd 20 >> $out/ab.log 2>&1
RN_CONV_SHAPES="128,8,8,640,640,3" RN_CONV_EP=$ep RN_CONV_VARIANT="0/131072,0/0,0/131074" timeout -k 10 200 python tools/conv_bench.py dgrad 20 >> $out/ab.log 2>&1
done
grep -v amdgpu.ids $out/ab.log
timeout -k 10 500 python bench.py --breakdown --per-op 100 --no-cpu-baseline --also= > $out/bench.json 2> $out/bench.err
cat $out/bench.json

#!/bin/bash
out=gpurun_out/r4f
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_production_tiles.py -x -q -m gpu -k "wgrad9 or igemm8r_split" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
RN_CONV_VARIANT="0/16396,0/32776,0/8" timeout -k 10 200 python tools/conv_bench.py wgrad 20 >> $out/ab.log 2>&1
RN_CONV_SHAPES="128,8,8,640,640,3" RN_CONV_VARIANT="0/131072,0/0" timeout -k 10 200 python tools/conv_bench.py fwd 20 >> $out/ab.log 2>&1
grep -v amdgpu.ids $out/ab.log
timeout -k 10 500 python bench.py --no-cpu-baseline --also= > $out/bench.json 2> $out/bench.err
cat $out/bench.json

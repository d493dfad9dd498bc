#!/bin/bash
out=gpurun_out/r4ac
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_production_tiles.py -x -q -m gpu -k "wgrad9 or wgrad8r" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "wrn28" > $out/tests2.log 2>&1 || { tail -40 $out/tests2.log; exit 1; }
tail -2 $out/tests2.log
RN_CONV_VARIANT="0/16396,0/32776,0/8,0/264,0/520,0/776" timeout -k 10 300 python tools/conv_bench.py wgrad 20 > $out/ab.log 2>&1; grep -v amdgpu.ids $out/ab.log

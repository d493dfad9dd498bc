#!/bin/bash
out=gpurun_out/r4j
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_production_tiles.py -x -q -m gpu -k "wgrad8r_batch or wgrad9" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "wrn28_10" > $out/tests2.log 2>&1 || { tail -40 $out/tests2.log; exit 1; }
tail -2 $out/tests2.log
timeout -k 10 500 python bench.py --no-cpu-baseline --also= > $out/bench.json 2> $out/bench.err
cat $out/bench.json | cut -c1-900

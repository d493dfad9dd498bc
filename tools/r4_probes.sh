#!/bin/bash
# timing probes of the round-4 kernels (wrong results, timing only): full / no LDS-DMA / no MFMA / no fragment reads
out=gpurun_out/r4b
mkdir -p $out
RN_CONV_VARIANT="0/2,0/18,0/34,0/50" timeout -k 10 200 python tools/conv_bench.py fwd 20 > $out/probes.log 2>&1
RN_CONV_VARIANT="0/8,0/264,0/520,0/776" timeout -k 10 200 python tools/conv_bench.py wgrad 20 >> $out/probes.log 2>&1
cat $out/probes.log

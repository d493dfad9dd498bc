#!/bin/bash
out=gpurun_out/r4v
mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-parity --also= $* 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_spread']['median'])"; }
for i in 1 2 3; do echo "shipped" >> $out/env.log; run >> $out/env.log; echo "RN_BN_APPLY_ROWS=old" >> $out/env.log; RN_BN_APPLY_ROWS=old run >> $out/env.log; done
cat $out/env.log
python bench.py --no-cpu-baseline --no-parity --also= --breakdown 2>&1 >/dev/null | grep -E "BN_APPLY|BN_BWD_APPLY"
RN_BN_APPLY_ROWS=old python bench.py --no-cpu-baseline --no-parity --also= --breakdown 2>&1 >/dev/null | grep -E "BN_APPLY|BN_BWD_APPLY"
for wl in v2-164 rn20 wrn-50-2b; do echo "$wl"; run --workload $wl --steps 10; RN_BN_APPLY_ROWS=old run --workload $wl --steps 10; done
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bn" 2>&1 | tail -2

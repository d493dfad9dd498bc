#!/bin/bash
# Same-box A/B of two ENVIRONMENT settings of one build (lowering switches, RN_VARIANT kernel switches): boxes differ by 3-8 %, so the two settings
# alternate on one box and the pairs are read, not the means.
#   gpurun -- 'bash tools/ab_env.sh 3 wrn-50-2b "RN_POOL_GATHER_SUMS=1 RN_POOL_NO_COLSUM=1 RN_VARIANT=65536" ""'
# arguments: rounds, workload, setting A, setting B (each a space-separated list of NAME=value, may be empty), optional dtype (fp16)
set -e
rounds=${1:-2}; wl=${2:-wrn-28-10}; A=$3; B=$4; dt=${5:-fp16}
run() { env $1 python bench.py --workload $wl --dtype $dt --no-cpu-baseline --no-parity --also= | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in $(seq $rounds); do
  echo "round $i A [$A] $wl $dt $(run "$A") ms/step"
  echo "round $i B [$B] $wl $dt $(run "$B") ms/step"
done

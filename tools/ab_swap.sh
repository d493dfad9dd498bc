#!/bin/bash
# Same-box A/B of two builds of librn_hip.so (boxes differ by 3-8 %, so two runs on two boxes say nothing).
#   here:        git stash; make -C pytorch_ddp_resnet_amd/csrc; cp pytorch_ddp_resnet_amd/librn_hip.so ab_old.so; git stash pop
#                make -C pytorch_ddp_resnet_amd/csrc; cp pytorch_ddp_resnet_amd/librn_hip.so ab_new.so
#   on the box:  gpurun -- 'bash tools/ab_swap.sh 3 wrn-28-10 v2-164'      (rounds, workloads ...)
# The two libraries are swapped in turn under the package path, each followed by one bench run per workload; read the pairs, not the means.
set -e
rounds=${1:-2}; shift || true
wls=${@:-wrn-28-10}
for i in $(seq $rounds); do
  for v in old new; do
    cp ab_$v.so pytorch_ddp_resnet_amd/librn_hip.so
    for w in $wls; do
      ms=$(python bench.py --workload $w --dtype fp16 --no-cpu-baseline --no-parity --also= 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "round $i $v $w $ms ms/step"
    done
  done
done
cp ab_new.so pytorch_ddp_resnet_amd/librn_hip.so

#!/bin/bash
# Same-box A/B of two builds of librn_hip.so (boxes differ by 3-8 %, so two runs on two boxes say nothing).
#   here:        git stash; make -C pytorch_ddp_resnet_amd/csrc; cp pytorch_ddp_resnet_amd/librn_hip.so ab_old.so; git stash pop
#                make -C pytorch_ddp_resnet_amd/csrc; cp pytorch_ddp_resnet_amd/librn_hip.so ab_new.so
#   on the box:  gpurun -- 'bash tools/ab_swap.sh 3 wrn-28-10 v2-164'      (rounds, workloads ...)
# The two libraries are swapped in turn under the package path, each followed by one bench run per workload; read the pairs, not the means.
set -e
trap 'cp ab_new.so pytorch_ddp_resnet_amd/librn_hip.so' EXIT          # whatever happens, the package ends with the NEW library
rounds=${1:-2}; shift || true
wls=${@:-wrn-28-10}
ver() { python -c "import ctypes,sys; print(ctypes.CDLL(sys.argv[1]).rn_version())" "$1"; }
if [ "$(ver ./ab_old.so)" != "$(ver ./ab_new.so)" ]; then
  echo "ab_swap: the two libraries report different ABI versions ($(ver ./ab_old.so) vs $(ver ./ab_new.so)): the Python binding fits only one of them" >&2
  exit 1
fi
for i in $(seq $rounds); do
  for v in old new; do
    cp ab_$v.so pytorch_ddp_resnet_amd/librn_hip.so
    for w in $wls; do
      ms=$(python bench.py --workload $w --dtype fp16 --no-cpu-baseline --no-parity --also= | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
      echo "round $i $v $w $ms ms/step"
    done
  done
done
cp ab_new.so pytorch_ddp_resnet_amd/librn_hip.so

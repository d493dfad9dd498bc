#!/bin/bash
# One round's profiles on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh <tag> <workload> <dtype> [bench args]
# kernel trace + stats of the bench command, then the three PMC passes separately (SQ, FETCH_SIZE, WRITE_SIZE), as
# MI355X_MICROARCH.md prescribes.  The program itself follows `--` (no env/bash hop under rocprofv3).  The PMC passes keep 8 warm-up steps: the fp16
# loss-scale backoff of bench.py acts during warm-up only, and a step full of inf / nan gradients is not the step being profiled (bench.py also
# reports `grads_finite`).
set -e
tag=$1; wl=$2; dt=$3; shift 3
out=gpurun_out/prof_${tag}
mkdir -p $out
export TMPDIR=/tmp
B="bench.py --workload $wl --dtype $dt --no-cpu-baseline --no-parity --also= $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $B --steps 20 --warmup 5 > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o t -- python3 $B --steps 2 --warmup 8 > $out/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o t -- python3 $B --steps 2 --warmup 8 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o t -- python3 $B --steps 2 --warmup 8 > $out/write.log 2>&1
find $out -name "*.csv" | head -20

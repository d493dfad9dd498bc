#!/bin/bash
# the same kernel hot (tools/conv_bench.py, the MODEL's operand sets) under the PMC passes, to set beside the step's passes
# (tools/profile_round.sh).  Counters in passes of their own; the program itself follows `--`; the environment is exported beforehand.
set -e
out=gpurun_out/pmc_gap
mkdir -p $out
export TMPDIR=/tmp RN_CONV_EP=1 RN_CONV_VARIANT=0/0 RN_CONV_ROUNDS=1
for which in fwd dgrad; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/${which}_sq -o t -- python3 tools/conv_bench.py $which 5 > $out/${which}_sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${which}_fetch -o t -- python3 tools/conv_bench.py $which 5 > $out/${which}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${which}_write -o t -- python3 tools/conv_bench.py $which 5 > $out/${which}_write.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${which}_trace -o t -- python3 tools/conv_bench.py $which 5 > $out/${which}_trace.log 2>&1
done
find $out -name "*.csv" | head -30

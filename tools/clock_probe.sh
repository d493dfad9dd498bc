#!/bin/bash
# does the chip hold its clock under the bench's load?  rocm-smi samples (clocks, power) while bench.py runs 300 steps
out=gpurun_out/clock
mkdir -p $out
(python bench.py --no-cpu-baseline --no-parity --also= --steps 300 --warmup 5 > $out/bench.json 2>/dev/null) &
BP=$!
sleep 12
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' >> $out/smi.log
  echo >> $out/smi.log
  sleep 0.2
  kill -0 $BP 2>/dev/null || break
done
wait $BP
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $out/smi_idle.log
cut -c1-400 $out/bench.json
echo
head -30 $out/smi.log | cut -c1-300
echo idle; cat $out/smi_idle.log | cut -c1-300

#!/bin/bash
python scripts/drift_probe.py 60 > gpurun_out/drift.log 2>&1 &
PID=$!
for i in $(seq 1 40); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power \(W\)|sclk|mclk|fclk|Temperature" | sed 's/GPU\[0\]//g' | tr '\n' ' ' | sed 's/  */ /g; s/clock level: //g; s/Temperature (Sensor //g'; echo
  sleep 0.8
done > gpurun_out/drift_smi.log
wait $PID
grep "us/eval" gpurun_out/drift.log | awk 'NR%3==1' 
awk 'NR%4==1' gpurun_out/drift_smi.log | cut -c1-330

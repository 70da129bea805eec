#!/bin/bash
# Samples clocks / power with rocm-smi while the fused kernel runs in a loop with different ablation bits
# (is the store stream slowed by a power-management clock drop when the matrix work is on?).
ROOT=$(pwd)
mkdir -p gpurun_out
for AB in 0 4 2 41; do
  python scripts/tune.py --batch 8 --ablate $AB --steps 12000 > gpurun_out/pp_$AB.log 2>&1 &
  PID=$!
  sleep 6
  for i in 1 2 3 4 5 6; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk" | tr '\n' ' ' ; echo
    sleep 0.7
  done > gpurun_out/pp_smi_$AB.log
  wait $PID
  echo "== ablate $AB"; grep -o '"us_per_eval": [0-9.]*' gpurun_out/pp_$AB.log; cat gpurun_out/pp_smi_$AB.log | sed 's/GPU\[0\]//g; s/\t/ /g; s/  */ /g' | cut -c1-400
done

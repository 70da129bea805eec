#!/bin/bash
# stagger experiment: the column-group Hessian kernel with its waves' starts spread over the SIMDs / wave slots
for rnd in 1 2; do
for o in "" "-DHC_STAGGER=8" "-DHC_STAGGER=32" "-DHC_STAGGER=100"; do
  echo "== round $rnd opts '$o'"
  PCL_JIT_OPTS="$o" HC_ONLY=1 python lab/probes/hess_cols_ab.py 8,10 8,64 2>&1 | grep -v amdgpu.ids
done
done

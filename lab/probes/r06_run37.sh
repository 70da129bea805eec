#!/bin/bash
# occupancy experiment: the column-group Hessian kernel with its LDS request padded (fewer waves per CU)
for rnd in 1 2; do
for pad in 0 2048 6000 12000 20000; do
  echo "== round $rnd LDS pad $pad"
  PCL_HC_LDS_PAD=$pad HC_ONLY=1 python lab/probes/hess_cols_ab.py 8,10 8,64 2>&1 | grep -v amdgpu.ids
done
done

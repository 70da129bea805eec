#!/bin/bash
# rounds of wave slots: does a LOWER occupancy (padded LDS request) that balances the rounds beat the greedy fill?  order 10, 3 / 6 / 7 / 8 trajectories per launch
for pad in 0 3600 8000 13500; do
  echo "== LDS pad $pad"
  PCL_HC_LDS_PAD=$pad HC_ONLY=1 python lab/probes/hess_cols_ab.py 10 3,6,7,8 2>&1 | grep -v amdgpu.ids
done

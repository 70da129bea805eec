#!/bin/bash
# same-box A/B: the tree as committed (gpurun_ab/r06_old: R tiles staged through registers, q - 2 of them in LDS) against the working tree (one tile, copied by the memory pipe)
for rnd in 1 2 3; do
  echo "== round $rnd old"
  (cd gpurun_ab/r06_old && HC_ONLY=1 python lab/probes/hess_cols_ab.py 8,10 8,64 2>&1 | grep -v amdgpu.ids)
  echo "== round $rnd new"
  HC_ONLY=1 python lab/probes/hess_cols_ab.py 8,10 8,64 2>&1 | grep -v amdgpu.ids
done

#!/bin/bash
# rocprofv3 profiles of the Hessian-of-the-Lagrangian and residual-only kernels at BASELINE config 3, 8 trajectories per launch:
# kernel trace + separate counter passes (never combined).  Writes gpurun_out/profiles_<tag>/<tag>_hess_summary.json and
# <tag>_eval_summary.json (copy into profiles/).   Usage (on the GPU box): scripts/profile_hess_eval.sh r02
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_he_$TAG
mkdir -p $OUT $ROOT/gpurun_out/profiles_$TAG
export TMPDIR=/tmp
cd /tmp
for W in hess:4 hess:3 eval:2 eval:1; do
  what=${W%%:*}; kern=${W##*:}
  script=$ROOT/lab/probes/${what}_sparse_run.py
  name=${what}_k${kern}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -o $name -- python $script 8 $kern > /dev/null 2>&1
  i=0
  for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "WRITE_SIZE" "FETCH_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $P --output-format csv -d $OUT/${name}_pmc$i -o $name -- python $script 8 $kern > /dev/null 2>&1
  done
done
cd $ROOT
python scripts/summarize_hess_eval.py $OUT gpurun_out/profiles_$TAG $TAG

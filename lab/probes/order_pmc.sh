#!/bin/bash
# counters of the lock-step general-order kernel (order 10, config 3, one trajectory, compact output)
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_order
mkdir -p $OUT
cd /tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
         "SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -o o -- python $ROOT/lab/probes/order_run.py 10 1 1 > /dev/null 2>&1
  python $ROOT/lab/probes/pmc_sum.py $OUT/pmc$i pcl_pade_v2 2>&1 | tail -12
done

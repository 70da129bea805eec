#!/bin/bash
# Profile the bench workload with rocprofv3 on the GPU box; raw output under gpurun_out/prof_<tag>/,
# summaries (to be committed) under gpurun_out/profiles_<tag>/ -> copy into profiles/.
# Usage: scripts/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:-"--steps 30 --warmup 5 --no-cpu-baseline --no-shares --no-extras"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/bench.py $ARGS > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run trace --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run pmc_sq2 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
cd $ROOT
python scripts/summarize_profile.py $OUT gpurun_out/profiles_$TAG $TAG

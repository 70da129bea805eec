#!/bin/bash
# rocprofv3 kernel trace + SQ counters for the Hessian kernel (scripts/hess_bench.py); output under gpurun_out/prof_hess/
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_hess
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/scripts/hess_bench.py 0 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run trace --kernel-trace --stats
run pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run pmc_sq2 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
run pmc_sq3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_WAVE32_INSTS
cd $ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof_hess/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "hess" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
for f in glob.glob("gpurun_out/prof_hess/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY

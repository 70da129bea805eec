#!/bin/bash
# counter passes of the column-group Hessian kernel (8 trajectories per launch, order 8) -> gpurun_out/hc_pmc/summary.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/hc_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o hc -- python $ROOT/lab/probes/hess_cols_run.py ${1:-8} ${2:-8} ${3:-8} > /dev/null 2>&1
done
cd $ROOT
python - <<'PY' > gpurun_out/hc_pmc/summary.txt
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('gpurun_out/hc_pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hess' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot): print(k, tot[k] / max(n[k], 1), n[k])
PY
cat gpurun_out/hc_pmc/summary.txt

#!/bin/bash
# rocprofv3 passes of the Hessian of the Lagrangian on ONE trajectory per launch (config 3, orders 8 and 10): the two-wave kernel (pcl_hess_cols_pair_kernel,
# `auto`) and the one-wave kernel it replaces there (hess_pair=0), kernel trace and two counter passes each (never in one run with the trace).
# Usage: scripts/profile_hess_single.sh <tag>   ->  gpurun_out/profiles_<tag>/<tag>_hess_single.json (+ the kernel_stats csv files); copy into profiles/.
set -u
TAG=${1:-r06}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_hs
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT $DST
export TMPDIR=/tmp
cd /tmp
runpy() { name=$1; args=$2; shift 2; rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/lab/probes/hess_cols_run.py $args > $OUT/$name.log 2> $OUT/$name.err; echo "$name rc=$?"; }
for order in 8 10; do
  for pair in 1 0; do
    n=o${order}_p${pair}
    runpy trace_$n "1 8 $order hess_pair=$pair launches=60" --kernel-trace --stats
    runpy sq1_$n "1 8 $order hess_pair=$pair" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
    runpy sq2_$n "1 8 $order hess_pair=$pair" --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU
  done
done
cd $ROOT
python - $OUT $DST $TAG <<'PY'
import csv, glob, json, sys, collections, shutil
out, dst, tag = sys.argv[1:4]
res = {"workload": "Hessian of the Lagrangian, config 3, ONE trajectory per launch (lab/probes/hess_cols_run.py 1 8 <order> hess_pair=<0|1>)", "runs": {}}
for order in (8, 10):
    for pair in (1, 0):
        n = "o%d_p%d" % (order, pair)
        r_ = {"kernels": {}, "counters_per_dispatch": {}}
        for f in glob.glob("%s/trace_%s/**/*kernel_stats.csv" % (out, n), recursive=True):
            for r in csv.DictReader(open(f)):
                if "hess" in r["Name"]:
                    r_["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
            shutil.copy(f, "%s/%s_hess_single_%s_kernel_stats.csv" % (dst, tag, n))
        tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
        for p in ("sq1", "sq2"):
            for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, p, n), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "hess_cols" in r["Kernel_Name"]:
                        tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        for k in sorted(tot): r_["counters_per_dispatch"][k] = tot[k] / cnt[k]
        res["runs"]["order %d, %s" % (order, "chain wave + contribution wave (auto)" if pair else "one wave per column group (hess_pair = 0)")] = r_
        print(n, json.dumps(r_["kernels"]))
json.dump(res, open("%s/%s_hess_single.json" % (dst, tag), "w"), indent=1)
PY

#!/bin/bash
# Profile the bench workloads with rocprofv3 on the GPU box; raw output under gpurun_out/prof_<tag>/, summaries (to be
# committed) under gpurun_out/profiles_<tag>/ -> copy into profiles/.   Usage: scripts/profile.sh <tag>
# Counter passes never share a run with --kernel-trace / --stats (separate runs, as the MI355X guide prescribes).
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { name=$1; args=$2; shift 2; rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/bench.py $args > $OUT/$name.log 2> $OUT/$name.err; echo "$name rc=$?"; }
ALL="--steps 50 --warmup 5 --no-cpu-baseline --no-resident"                                   # headline + shares + the other rates: every kernel of the path
SINGLE="--workload single --steps 50 --warmup 5 --no-cpu-baseline --no-extras"   # the headline launch only
MULTI="--workload multistart --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
run trace "$ALL" --kernel-trace --stats
run trace_single "$SINGLE" --kernel-trace --stats     # the headline launch alone: its average duration, unmixed with the compact launches of the same grid
run trace_multistart "$MULTI" --kernel-trace --stats
run single_write "$SINGLE" --pmc WRITE_SIZE
run single_fetch "$SINGLE" --pmc FETCH_SIZE
run multistart_write "$MULTI" --pmc WRITE_SIZE
run multistart_fetch "$MULTI" --pmc FETCH_SIZE
run all_write "$ALL" --pmc WRITE_SIZE
run all_sq1 "$ALL" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run all_sq2 "$ALL" --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
run single_tcc "$SINGLE" --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
run single_tcc2 "$SINGLE" --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_64B_sum
run multistart_tcc2 "$MULTI" --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
PCL_V4_TICKET=0 run multistart_static_tcc2 "$MULTI" --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum   # the static split beside the slice tickets
PCL_V4_TICKET=0 run trace_multistart_static "$MULTI" --kernel-trace --stats
# the order that matches the reference's exp constraint at config 3, with passes of its own
run trace_single_order8 "$SINGLE --order 8 --warmup 600" --kernel-trace --stats   # (600 untimed launches: at orders 8 and 10 the clocks follow the load for several hundred launches)
run single_order8_write "$SINGLE --order 8" --pmc WRITE_SIZE
run single_order8_sq "$SINGLE --order 8" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
# ... and order 10, what the default constructor (pade_order = 0, tol 1e-10) picks on config 3's bounds since round 5
run trace_single_order10 "$SINGLE --order 10 --warmup 600" --kernel-trace --stats
run single_order10_write "$SINGLE --order 10" --pmc WRITE_SIZE
# the matrix-core kernel (kernel_version 3) beside the benchmarked one: MFMA instruction count and busy cycles per dispatch
run trace_single_k3 "$SINGLE --kernel-version 3" --kernel-trace --stats
run single_k3_mfma "$SINGLE --kernel-version 3" --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU
run single_mfma "$SINGLE" --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU
# the Hessian of the Lagrangian at order 8, 8 trajectories per launch: the column-group kernel and the kernel it replaced
runpy() { name=$1; args=$2; shift 2; rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python $ROOT/lab/probes/hess_cols_run.py $args > $OUT/$name.log 2> $OUT/$name.err; echo "$name rc=$?"; }
runpy trace_hess8 "8 8 8" --kernel-trace --stats
runpy trace_hess8_chains "8 7 8" --kernel-trace --stats
runpy hess8_sq1 "8 8 8" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
runpy hess8_sq2 "8 8 8" --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU
runpy hess8_write "8 8 8" --pmc WRITE_SIZE
runpy hess8_fetch "8 8 8" --pmc FETCH_SIZE
# ... and at order 10, the order the default constructor picks on config 3's bounds (round-5 review)
runpy trace_hess10 "8 8 10" --kernel-trace --stats
runpy hess10_sq1 "8 8 10" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
runpy hess10_sq2 "8 8 10" --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU
runpy hess10_write "8 8 10" --pmc WRITE_SIZE
runpy hess10_fetch "8 8 10" --pmc FETCH_SIZE
cd $ROOT
python scripts/summarize_profile.py $OUT gpurun_out/profiles_$TAG $TAG
python - $OUT gpurun_out/profiles_$TAG $TAG <<'PY'
# <tag>_hess_order8.json, <tag>_hess_order10.json: per-dispatch averages of the general-order Hessian kernels (trace) and of the counter passes
import csv, glob, json, sys, collections, shutil
out, dst, tag = sys.argv[1:4]
for order in (8, 10):
    res = {"workload": "Hessian of the Lagrangian, config 3, order %d, 8 trajectories per launch (lab/probes/hess_cols_run.py)" % order, "kernels": {}, "counters_per_dispatch": {}}
    for name in ("trace_hess%d" % order,) + (("trace_hess8_chains",) if order == 8 else ()):
        for f in glob.glob("%s/%s/**/*kernel_stats.csv" % (out, name), recursive=True):
            for r in csv.DictReader(open(f)):
                if "hess" in r["Name"]:
                    res["kernels"][r["Name"] + (" (hess_kernel 7)" if name.endswith("chains") else "")] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
            shutil.copy(f, "%s/%s_%s_kernel_stats.csv" % (dst, tag, name.replace("trace_", "")))
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for name in ("hess%d_sq1" % order, "hess%d_sq2" % order, "hess%d_write" % order, "hess%d_fetch" % order):
        for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, name), recursive=True):
            for r in csv.DictReader(open(f)):
                if "hess_cols" in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(tot): res["counters_per_dispatch"][k] = tot[k] / n[k]
    c = res["counters_per_dispatch"]
    if "WRITE_SIZE" in c: res["write_MB_per_dispatch"] = c["WRITE_SIZE"] * 1024 / 1e6  # (KiB)
    if "FETCH_SIZE" in c: res["fetch_MB_per_dispatch"] = 2 * c["FETCH_SIZE"] * 1024 / 1e6  # (KiB, tallied at half the line size on gfx950: doubled)
    res["algorithmic_MB_per_dispatch"] = 8 * 99 * 20440 * 8 / 1e6
    json.dump(res, open("%s/%s_hess_order%d.json" % (dst, tag, order), "w"), indent=1)
    print(order, json.dumps(res["kernels"]))
PY

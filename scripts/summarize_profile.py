#!/usr/bin/env python3
"""Condense rocprofv3 output of scripts/profile.sh (one kernel-trace run + separate PMC runs) into small committed summaries:
<tag>_kernel_stats.csv, <tag>_summary.json, pmc_traffic.json (HBM bytes per launch of the fused kernel, per bench workload)."""
import csv, glob, json, os, sys
from collections import defaultdict

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
summary = {"tag": tag}


def find(run, pat):
    return sorted(glob.glob(os.path.join(src, run, "**", pat), recursive=True))


def short(name):
    return name.split("(")[0][:90]


for f in find("trace", "*kernel_stats.csv"):
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as o:
        o.write(open(f).read())
for f in find("trace", "*kernel_trace.csv"):
    dur, meta = defaultdict(list), {}
    for r in csv.DictReader(open(f)):
        name = short(r.get("Kernel_Name", "?"))
        key = "%s | grid %s" % (name, r.get("Grid_Size_X", r.get("Grid_Size")))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[key] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")}
    summary["kernel_trace"] = {k: dict(calls=len(v), avg_us=sum(v) / len(v) / 1e3, min_us=min(v) / 1e3, max_us=max(v) / 1e3, **meta[k])
                               for k, v in dur.items() if "pcl_" in k}
for wl in ("single", "multistart", "multistart_static", "single_order8", "single_order10", "single_k3"):  # the headline launches alone (the full trace mixes them with compact launches of the same grid)
    for f in find("trace_" + wl, "*kernel_trace.csv"):
        dur = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pcl_fused" in r.get("Kernel_Name", ""):
                dur["%s | grid %s" % (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        # (avg_last50_us: the timed launches of the pass -- the passes at orders 8 and 10 run 600 untimed launches first, the clocks follow the load slowly;
        #  the trace file lists dispatches in the order they ran)
        summary["kernel_trace_" + wl] = {k: dict(calls=len(v), avg_us=sum(v) / len(v) / 1e3, min_us=min(v) / 1e3, max_us=max(v) / 1e3,
                                                 avg_last50_us=sum(v[-50:]) / len(v[-50:]) / 1e3) for k, v in dur.items()}
    for f in find("trace_" + wl, "*kernel_stats.csv"):
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)), "w") as o:
            o.write(open(f).read())
pmc = {}
for run in sorted(os.listdir(src)):
    if not os.path.isdir(os.path.join(src, run)) or run.startswith("trace"):
        continue
    for f in find(run, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "pcl_" not in r["Kernel_Name"]:
                continue
            acc["%s | grid %s" % (short(r["Kernel_Name"]), r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kname, cs in acc.items():
            for cname, vals in cs.items():
                pmc.setdefault(run, {}).setdefault(kname, {})[cname] = dict(n=len(vals), avg=sum(vals) / len(vals), min=min(vals), max=max(vals))
summary["pmc_per_dispatch"] = pmc
# HBM traffic of the fused kernel per workload (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE in KiB from separate
# --pmc passes; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x -> doubled before use.
traffic = {}
for wl, units in (("single", 1), ("multistart", 8)):
    w, r = pmc.get(wl + "_write", {}), pmc.get(wl + "_fetch", {})
    kw = [k for k in w if ("pcl_fused_kernel_v3" in k or "pcl_fused_sparse_kernel" in k) and "WRITE_SIZE" in w[k]]
    kr = [k for k in r if ("pcl_fused_kernel_v3" in k or "pcl_fused_sparse_kernel" in k) and "FETCH_SIZE" in r[k]]
    if kw and kr:
        wr, rd = w[kw[0]]["WRITE_SIZE"]["avg"] * 1024.0, r[kr[0]]["FETCH_SIZE"]["avg"] * 1024.0
        traffic[wl] = dict(workload=wl, batch=units, knots=100, kernel=kw[0], write_bytes=wr, fetch_bytes_raw=rd, fetch_bytes_corrected=2.0 * rd,
                           hbm_bytes_per_launch=wr + 2.0 * rd, source="rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes), " + tag)
summary["hbm_traffic"] = traffic
# MFMA A/B: the benchmarked kernel against the matrix-core kernel (kernel_version 3), per dispatch.  Utilisation = MFMA busy cycles / (SIMDs x
# kernel time x clock): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs; 256 CUs x 4 SIMDs at 2.4 GHz.
ab = {}
for tagk, run, tr in (("benchmarked_kernel", "single_mfma", "kernel_trace_single"), ("kernel3_matrix_cores", "single_k3_mfma", "kernel_trace_single_k3")):
    for kname, cs in pmc.get(run, {}).items():
        if "pcl_fused" not in kname:
            continue
        t_us = next((v["avg_us"] for k, v in summary.get(tr, {}).items() if k == kname), None)
        busy = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("avg", 0.0)
        ab[tagk] = dict(kernel=kname, us_per_launch=t_us, mfma_mops_f64=cs.get("SQ_INSTS_VALU_MFMA_MOPS_F64", {}).get("avg"), mfma_busy_cycles=busy,
                        valu_insts=cs.get("SQ_INSTS_VALU", {}).get("avg"),
                        mfma_util=(busy / (256 * 4 * t_us * 1e-6 * 2.4e9)) if t_us else None, source="rocprofv3 --pmc (own pass), " + tag)
if ab:
    json.dump(ab, open(os.path.join(dst, "mfma_ab.json"), "w"), indent=1)
    summary["mfma_ab"] = ab
if traffic:
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(src, "trace.log")):
    for line in open(f, errors="ignore"):
        if line.startswith('{"metric"'):
            summary["bench_line_under_profiler"] = json.loads(line)
json.dump(summary, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1)
for k, v in sorted(summary.get("kernel_trace", {}).items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"]):
    print("%-100s calls=%4d avg=%9.2f us (min %.2f max %.2f)" % (k[:100], v["calls"], v["avg_us"], v["min_us"], v["max_us"]))
for wl in ("single", "multistart", "multistart_static", "single_order8", "single_order10"):
    for k, v in summary.get("kernel_trace_" + wl, {}).items():
        print("%-11s alone: %-70s calls=%4d avg=%9.2f us (min %.2f max %.2f; last 50: %.2f)" % (wl, k[:70], v["calls"], v["avg_us"], v["min_us"], v["max_us"], v["avg_last50_us"]))
for wl, t in traffic.items():
    print(wl, "HBM bytes per launch %.4g (write %.4g, fetch corrected %.4g)" % (t["hbm_bytes_per_launch"], t["write_bytes"], t["fetch_bytes_corrected"]))

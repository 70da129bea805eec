#!/usr/bin/env python3
"""Summaries of scripts/profile_hess_eval.sh: per kernel average duration (kernel trace) and counters per dispatch."""
import csv, glob, json, os, sys, collections
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
want = ("hess", "eval", "sparse_values", "fused_kernel")
for what in ("hess", "eval"):
    out = {"tag": tag, "workload": "BASELINE config 3 (d=27, m=6, N=100), 8 trajectories per launch, %s" % ("Hessian of the Lagrangian" if what == "hess" else "residual only"),
           "note": "kernel 4 / eval_kernel 2 = pattern-compiled kernels (DESIGN 4.7), kernel 3 / eval_kernel 1 = matrix-core kernels; counters are per dispatch, separate passes",
           "variants": {}}
    for d in sorted(glob.glob(os.path.join(src, what + "_k*_trace"))):
        name = os.path.basename(d)[: -len("_trace")]
        var = {"kernel_trace": {}, "pmc_per_dispatch": {}}
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            per = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                per[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
            for k, v in per.items():
                if any(w in k for w in want):
                    v = v[2:] if len(v) > 4 else v  # (first launches: compilation, cold caches)
                    var["kernel_trace"][k[:80]] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
        for pd in sorted(glob.glob(os.path.join(src, name + "_pmc*"))):
            for f in glob.glob(pd + "/**/*counter_collection.csv", recursive=True):
                acc = collections.defaultdict(lambda: collections.defaultdict(list))
                for row in csv.DictReader(open(f)):
                    if any(w in row["Kernel_Name"] for w in want):
                        acc[row["Kernel_Name"][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
                for k, cs in acc.items():
                    for cn, v in cs.items():
                        var["pmc_per_dispatch"].setdefault(k, {})[cn] = sum(v) / len(v)
        # derived
        for k, cs in var["pmc_per_dispatch"].items():
            if "SQ_LDS_IDX_ACTIVE" in cs and cs["SQ_LDS_IDX_ACTIVE"]:
                cs["lds_bank_conflict_frac"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"]
            if "WRITE_SIZE" in cs:
                cs["hbm_write_bytes"] = cs["WRITE_SIZE"] * 1024.0  # WRITE_SIZE is reported in KiB
            if "FETCH_SIZE" in cs:
                cs["hbm_fetch_bytes_corrected"] = cs["FETCH_SIZE"] * 1024.0 * 2.0  # gfx950: FETCH_SIZE counts 64-byte requests as 32 (MI355X guide)
        out["variants"][name] = var
    json.dump(out, open(os.path.join(dst, "%s_%s_summary.json" % (tag, what)), "w"), indent=1)
    print(what, {n: {k: round(v["avg_us"], 2) for k, v in var["kernel_trace"].items()} for n, var in out["variants"].items()})

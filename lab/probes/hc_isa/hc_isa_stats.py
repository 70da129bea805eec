#!/usr/bin/env python3
"""Static ISA statistics of pcl_hess_cols_kernel per phase (no GPU): generate the module's source with the code generator as it stands in the tree,
compile it for gfx950 (-S) and count instructions between the `; hc_mark <name>` comments the kernel leaves at its phase boundaries.
usage: hc_isa_stats.py [q=4] [csrc dir = piccolo.jl_amd/csrc]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
q = int(sys.argv[1]) if len(sys.argv) > 1 else 4
csrc = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "piccolo.jl_amd", "csrc")
tmp = os.environ.get("HC_TMP", "/tmp/hc_isa")
os.makedirs(tmp, exist_ok=True)
exe = os.path.join(tmp, "hc_dump")
subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", csrc, "-o", exe, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hc_dump.cpp")])
src = os.path.join(tmp, "hc_q%d.hip" % q)
with open(src, "w") as f:
    subprocess.check_call([exe, os.path.join(ROOT, "bench", "config3_inputs.bin"), str(q)], stdout=f)
asm = os.path.join(tmp, "hc_q%d.s" % q)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-include", "hip/hip_runtime.h", "-I", csrc, "--cuda-device-only", "-DHC_MARKS", "-S", "-o", asm, src] + os.environ.get("HC_FLAGS", "").split(),
                      stderr=subprocess.DEVNULL)
txt = open(asm).read()
body = txt[txt.index("pcl_hess_cols_kernel:"):]
body = body[:body.index("s_endpgm")]
meta = {k: re.search(r"\.%s:\s+(\d+)" % k, txt) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
print("q=%d " % q + " ".join("%s=%s" % (k, v.group(1) if v else "?") for k, v in meta.items()))
phase, stats = "head", collections.OrderedDict()
for ln in body.splitlines():
    mm = re.search(r";\s*hc_mark (\w+)", ln)
    if mm:
        phase = mm.group(1)
        continue
    t = ln.strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    c = stats.setdefault(phase, collections.Counter())
    kind = ("lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "scratch" if op.startswith("scratch_") else
            "wait" if op == "s_waitcnt" else "salu" if op.startswith("s_") else "mov" if op.startswith(("v_mov_b64", "v_mov_b32_e", "v_accvgpr")) else
            "f64" if re.match(r"v_(fma|fmac|mul|add)_f64", op) else "valu")
    c[kind] += 1
tot = collections.Counter()
print("%-14s %6s %6s %6s %6s %6s %6s %6s %6s" % ("phase", "f64", "valu", "mov", "lds", "vmem", "scratch", "wait", "salu"))
for ph, c in stats.items():
    print("%-14s %6d %6d %6d %6d %6d %6d %6d %6d" % (ph, c["f64"], c["valu"], c["mov"], c["lds"], c["vmem"], c["scratch"], c["wait"], c["salu"]))
    tot.update(c)
print("%-14s %6d %6d %6d %6d %6d %6d %6d %6d" % ("static total", tot["f64"], tot["valu"], tot["mov"], tot["lds"], tot["vmem"], tot["scratch"], tot["wait"], tot["salu"]))
loop = [ph for ph in stats if ph.startswith("pass_")]
pv = sum(stats[p]["f64"] + stats[p]["valu"] + stats[p]["mov"] for p in loop)
pl = sum(stats[p]["lds"] for p in loop)
rc = stats.get("rchain", collections.Counter())
print("per pass: %d vector, %d LDS | R chain step: %d vector, %d LDS | estimate per wave (q+1 passes, q-2 R steps): %d vector, %d LDS" % (
    pv, pl, rc["f64"] + rc["valu"] + rc["mov"], rc["lds"], pv * (q + 1) + (rc["f64"] + rc["valu"] + rc["mov"]) * max(q - 2, 0) + sum(
        stats[p]["f64"] + stats[p]["valu"] + stats[p]["mov"] for p in stats if not p.startswith("pass_") and p != "rchain"),
    pl * (q + 1) + rc["lds"] * max(q - 2, 0) + sum(stats[p]["lds"] for p in stats if not p.startswith("pass_") and p != "rchain")))

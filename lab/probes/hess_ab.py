#!/usr/bin/env python3
"""A/B of versions of the pattern-compiled Hessian kernel header (put them under lab/probes/ab/<name>.hpp) on one box: each version runs in its
own process (the generated module is cached per process), alternating, 8 and 16 trajectories per launch."""
import os, subprocess, sys, shutil
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = os.path.join(root, "piccolo.jl_amd", "csrc", "pcl_kernel_hessian_sparse.hpp")
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
out = []
for batch in (8, 16):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
    c = ms.ctx
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
    hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    ts = []
    for rep in range(5):
        for _ in range(5): c.hess_dev(Zd, mu, hv)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): c.hess_dev(Zd, mu, hv)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 40)
    out.append("batch %%d: median %%.1f min %%.1f" %% (batch, sorted(ts)[2], min(ts)))
    ms.close()
print(" | ".join(out))
''' % root
orig = open(H).read()
try:
    for rnd in range(3):
        for v in sys.argv[1:]:
            shutil.copy(os.path.join(root, "scripts", "probes", "ab", v + ".hpp"), H)
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
            print("%-6s %s" % (v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
finally:
    open(H, "w").write(orig)

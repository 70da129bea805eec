#!/usr/bin/env python3
"""The ensemble step's fused launch (8 members of config 4, per-member drifts, residual + Jacobian + reduce payload): the writer wave of
kernel 4 (auto) against the MERIT instance of kernel 3, and both against the plain launches; alternating in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
M, N = 8, 100
members = synthetic.config4_members(0, M)
traj = synthetic.synthetic_ensemble(members, N, seed=20260929 + 4)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    Bs = pa.BilinearIntegrator(members, traj, device=0, pade_order=4)
    c = Bs[0].ensemble.ctx
    c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(traj.datavec).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    ln, _ = c.merit_grad_len()
    out = torch.empty(ln, dtype=torch.float64, device="cuda")
    variants = {"v4 + payload": (0, True), "v3 + payload": (3, True), "v4 plain": (0, False), "v3 plain": (3, False)}
    res = {k: [] for k in variants}
    for rnd in range(7):
        for name in (list(variants) if rnd % 2 == 0 else list(variants)[::-1]):
            kv, merit = variants[name]
            c.set_option("kernel_version", kv)
            call = (lambda: c.eval_jac_merit_dev(Zd, None, dd, vd, out)) if merit else (lambda: c.eval_jac_dev(Zd, dd, vd))
            for _ in range(3):
                call()
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                call()
            e1.record(stream)
            stream.synchronize()
            if rnd:
                res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
            if rnd == 0:
                print(name, "kernel", c.get_option("last_kernel"), "fused", c.get_option("last_merit_fused"))
    for name, v in res.items():
        print("%-14s: %s  median %.1f us per 8-member launch (+ finish kernel where fused)" % (name, " ".join("%.1f" % x for x in v), np.median(v)), flush=True)

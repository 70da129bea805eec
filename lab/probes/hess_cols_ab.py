#!/usr/bin/env python3
"""General-order Hessian: the column-group kernel (hess_kernel 8: one wave = all chains of four state columns, a workgroup of its own) against
the chain-per-wave kernel (7, one / two workgroups per interval) and, at order 4, kernel 6.  1 / 8 / 64 trajectories per launch; the output
vectors bitwise, the scalar entries to rounding.  usage: hess_cols_ab.py [orders=4,8] [batches=1,8]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
orders = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4,8").split(",")]
batches = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "1,8").split(",")]
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for order in orders:
        for B in batches:
            trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
            t0 = trajs[0]
            Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
            c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            for kv in os.environ.get("HC_OPTS", "").split():  # e.g. HC_OPTS="hess_rpre=0"
                c.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
            hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
            variants = [("chains", 7, 0), ("columns", 8, 0)] + ([("chains/2", 7, 1)] if B == 1 else []) + ([("kernel 6", 0, 0)] if order == 4 else [])
            if os.environ.get("HC_ONLY"):  # the column-group kernel alone (A/B of its header variants: lab/probes/ab_jit_headers.sh)
                variants = [("columns", 8, 0)]
            res, outs = {v[0]: [] for v in variants}, {}
            for rnd in range(5):
                for name, hk, sp in (variants if rnd % 2 == 0 else variants[::-1]):
                    c.set_option("hess_kernel", hk)
                    c.set_option("hess_split", sp)
                    hv.fill_(float("nan"))
                    for _ in range(3):
                        c.hess_dev(Zd, mud, hv)
                    stream.synchronize()
                    outs[name] = hv.clone()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(20):
                        c.hess_dev(Zd, mud, hv)
                    e1.record(stream)
                    stream.synchronize()
                    res[name].append(e0.elapsed_time(e1) / 20 * 1e3)
            if os.environ.get("HC_ONLY"):
                print("order %2d B=%d: columns %.1f us (%.2f/eval) min %.1f, nan %d" % (order, B, np.median(res["columns"]), np.median(res["columns"]) / B, min(res["columns"]),
                                                                                         int(torch.isnan(outs["columns"]).sum().item())), flush=True)
                c.close()
                continue
            a, bb = outs["chains"].view(B * (t0.N - 1), -1), outs["columns"].view(B * (t0.N - 1), -1)
            nsc = (system.n_drives + 1) * (system.n_drives + 2) // 2
            scale = a.abs().max().item()
            print("order %2d B=%d: %s | columns vs chains: vectors bitwise %s (max diff %.1e), scalars max diff %.1e, nan %d" % (
                order, B, ", ".join("%s %.1f us (%.2f/eval)" % (k, np.median(v), np.median(v) / B) for k, v in res.items()),
                bool(torch.equal(a[:, nsc:], bb[:, nsc:])), (a[:, nsc:] - bb[:, nsc:]).abs().max().item() / scale,
                (a[:, :nsc] - bb[:, :nsc]).abs().max().item() / scale, int(torch.isnan(bb).sum().item())), flush=True)
            c.close()

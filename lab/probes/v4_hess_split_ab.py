#!/usr/bin/env python3
"""General-order Hessian kernel: one workgroup per interval against two (hess_split: half of the drive chains each, 512 registers per
lane), 1 / 2 / 8 trajectories per launch, orders 4 / 8 / 10; the order-4 kernel 6 beside them.  Bitwise equal; alternating in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for order in (4, 8, 10):
        for B in (1, 2, 8):
            trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
            t0 = trajs[0]
            Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
            c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            mud = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
            hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
            variants = [("one workgroup", 7, 0), ("two workgroups", 7, 1)] + ([("kernel 6", 0, 0)] if order == 4 else [])
            res, outs = {v[0]: [] for v in variants}, {}
            for rnd in range(5):
                for name, hk, sp in (variants if rnd % 2 == 0 else variants[::-1]):
                    c.set_option("hess_kernel", hk)
                    c.set_option("hess_split", sp)
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
            scale = outs["one workgroup"].abs().max().item()
            print("order %2d B=%d: %s | two vs one: bitwise %s, max diff %.1e%s" % (
                order, B, ", ".join("%s %.1f us" % (k, np.median(v)) for k, v in res.items()), bool(torch.equal(outs["one workgroup"], outs["two workgroups"])),
                (outs["one workgroup"] - outs["two workgroups"]).abs().max().item() / scale,
                (", kernel 6 vs one %.1e" % ((outs["kernel 6"] - outs["one workgroup"]).abs().max().item() / scale)) if order == 4 else ""), flush=True)
            c.close()

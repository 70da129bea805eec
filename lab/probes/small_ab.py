#!/usr/bin/env python3
"""Small systems (BASELINE config 2: CNOT, d = 4, N = 100; config 1: d = 2, N = 50): the one-wave-per-interval kernel (kernel_version 5,
auto) against kernel 1, residual + Jacobian and residual only, orders 4 and 8; HIP events, alternating in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    from piccolo_jl_amd.quantum import MultiTransmonSystem
    for cfg, N in ((2, 100), (1, 50), ("one 5-level transmon (d = 5)", 50), ("one 8-level transmon (d = 8)", 50), ("two transmons, 2 x 3 levels... (d = 6)", 50)):
        if cfg in (1, 2):
            system = synthetic.config_system(cfg)
        elif "d = 5" in cfg:
            system = MultiTransmonSystem([4.0], [0.2], [[0.0]], levels_per_transmon=5, drive_bounds=0.1)
        elif "d = 8" in cfg:
            system = MultiTransmonSystem([4.0], [0.2], [[0.0]], levels_per_transmon=8, drive_bounds=0.1)
        else:
            system = MultiTransmonSystem([4.0, 4.1], [0.2, 0.2], [[0, 0.1], [0.1, 0]], levels_per_transmon=[2, 3], drive_bounds=0.1) if False else MultiTransmonSystem([4.0], [0.2], [[0.0]], levels_per_transmon=6, drive_bounds=0.1)
        for order in (4, 8):
            t0 = synthetic.synthetic_trajectory(system, N, seed=7)
            Zd = torch.from_numpy(t0.datavec.copy()).cuda()
            ctxs = {}
            for name, kv in (("small", 0), ("kernel 1", 1)):
                if kv == 1 and order != 4:
                    continue
                c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                               dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                               Gj=system.G_drives_array(), batch=1, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
                c.set_stream(stream.cuda_stream)
                c.set_option("kernel_version", kv)
                ctxs[name] = c
            dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
            vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
            res = {}
            for rnd in range(6):
                for name, c in ctxs.items():
                    for what, call in (("R+J", lambda: c.eval_jac_dev(Zd, dd, vd)), ("R", lambda: c.eval_dev(Zd, dd))):
                        for _ in range(5):
                            call()
                        stream.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        for _ in range(200):
                            call()
                        e1.record(stream)
                        stream.synchronize()
                        res.setdefault((name, what), []).append(e0.elapsed_time(e1) / 200 * 1e3)
                        kid = c.get_option("last_kernel")
                        res.setdefault((name, what, "id"), kid)
            for key, v in res.items():
                if len(key) == 2:
                    print("config %s order %d %-9s %-4s: median %.2f us per launch (kernel id %d)" % (cfg, order, key[0], key[1], np.median(v), res[key + ("id",)]), flush=True)

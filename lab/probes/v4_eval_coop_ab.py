#!/usr/bin/env python3
"""Residual only, pattern-compiled: one wave per interval against four waves per interval (eval_coop), 1 / 2 / 4 / 8 trajectories per
launch, orders 4 and 8; bitwise equal, alternating in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for order in (4, 8):
        for B in (1, 2, 4, 8):
            trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
            t0 = trajs[0]
            Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
            c = pa.integrators._PclContext(d=system.levels, m=system.n_drives, N=t0.N, z_dim=t0.dim, u_off=t0.components["u"].start,
                                           dt_off=t0.components["Δt"].start, x_offs=[t0.components[pa.trajectory.STATE].start], G0=system.G_drift,
                                           Gj=system.G_drives_array(), batch=B, batch_mode=pa._lib.PCL_BATCH_TRAJ, pade_order=order)
            c.set_stream(stream.cuda_stream)
            dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
            res, outs = {0: [], 1: []}, {}
            for rnd in range(6):
                for coop in ((0, 1) if rnd % 2 == 0 else (1, 0)):
                    c.set_option("eval_coop", coop)
                    for _ in range(5):
                        c.eval_dev(Zd, dd)
                    stream.synchronize()
                    assert c.get_option("last_eval_coop") == coop and c.get_option("last_kernel") == 80 + order // 2
                    outs[coop] = dd.clone()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(100):
                        c.eval_dev(Zd, dd)
                    e1.record(stream)
                    stream.synchronize()
                    res[coop].append(e0.elapsed_time(e1) / 100 * 1e3)
            print("order %d B=%d: one wave per interval %.2f us/launch, four waves %.2f us/launch; bitwise equal: %s" %
                  (order, B, np.median(res[0]), np.median(res[1]), bool(torch.equal(outs[0], outs[1]))), flush=True)
            c.close()

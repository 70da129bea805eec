#!/usr/bin/env python3
"""Are the device-pointer entry points capturable in a HIP graph (torch.cuda.CUDAGraph), and what does replay buy for the
launch-bound cases (BASELINE config 2; one IPM iteration's worth of calls at config 3, one trajectory)?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic


def bench(fn, stream, reps=300):
    for _ in range(10):
        fn()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, (time.perf_counter() - t0) / reps * 1e6


for cfg in (2, 3):
    system = synthetic.config_system(cfg)
    traj = synthetic.synthetic_trajectory(system, 100, seed=20260929 + cfg)
    B = pa.HipPadeIntegrator(system.G_drift, system.G_drives_array(), traj, pade_order=4)
    c = B.ctx
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(traj.datavec).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")

        def iteration():  # what one interior-point iteration asks of this integrator
            c.eval_dev(Zd, dd)
            c.eval_jac_dev(Zd, dd, vd)
            c.hess_dev(Zd, mu, hv)

        iteration()
        stream.synchronize()
        ref = (dd.clone(), vd.clone(), hv.clone())
        ev, wall = bench(iteration, stream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            iteration()
        dd.zero_(); vd.zero_(); hv.zero_()
        g.replay()
        stream.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(ref, (dd, vd, hv)))
        evg, wallg = bench(g.replay, stream)
        print("config %d: eval + eval_jac + hess per iteration: %.1f us (events) / %.1f us (wall) direct; %.1f / %.1f us as a graph replay; identical results: %s"
              % (cfg, ev, wall, evg, wallg, same), flush=True)
    B.close()

#!/usr/bin/env python3
"""Residual only (pcl_eval_dev), config 3: one wave per interval against four waves per interval with two / one result tiles, 1 ... 16
trajectories per launch; the same bits.  usage: eval_coop_ab.py [order=4]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
for B in (1, 2, 4, 8, 10, 16):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, device=0, pade_order=order)
    c = ms.ctx
    c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    ref, line = None, []
    for mode in (0, 1, 2, -1):
        c.set_option("eval_coop", mode)
        dd.fill_(float("nan"))
        c.eval_dev(Zd, dd)
        c.sync()
        if ref is None:
            ref = dd.clone()
        assert torch.equal(dd, ref), (B, mode)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(50):
                c.eval_dev(Zd, dd)
            e1.record(stream)
            stream.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        line.append("%s %.2f us (%.2f/eval)" % ({0: "one wave", 1: "four waves, two tiles", 2: "four waves, one tile", -1: "auto"}[mode], np.median(ts), np.median(ts) / B))
    print("B=%2d order %d: " % (B, order) + " | ".join(line), flush=True)
    ms.close()

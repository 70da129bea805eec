#!/usr/bin/env python3
"""Column-group Hessian kernel (hess_kernel 8): option sets alternating in one process, bitwise check against the first one.
usage: hess_xcd_ab.py [order=8] [batches=1,8,64] [optset ...]   e.g. hess_xcd_ab.py 8 1,8,64 hess_xcd=0 hess_xcd=8"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
order = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "1,8,64").split(",")]
sets = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[3:]] or [{"hess_xcd": "0"}, {"hess_xcd": "8"}]
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
for B in batches:
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=order)
    c = ms.ctx
    c.set_stream(stream.cuda_stream)
    c.set_option("hess_kernel", 8)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
    hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
    ref = None
    for o in sets:
        for k, v in o.items():
            c.set_option(k, int(v))
        hv.fill_(float("nan"))
        c.hess_dev(Zd, mu, hv)
        c.sync()
        assert torch.isfinite(hv).all(), o
        if ref is None:
            ref = hv.clone()
        else:
            assert torch.equal(hv, ref), (o, float((hv - ref).abs().max()))
    reps = 20 if B <= 8 else 5
    res = [[] for _ in sets]
    for rnd in range(6):
        idx = list(range(len(sets)))
        for i in (idx if rnd % 2 == 0 else idx[::-1]):
            for k, v in sets[i].items():
                c.set_option(k, int(v))
            for _ in range(3):
                c.hess_dev(Zd, mu, hv)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                c.hess_dev(Zd, mu, hv)
            e1.record(stream)
            stream.synchronize()
            res[i].append(e0.elapsed_time(e1) / reps * 1e3)
    for o, r in zip(sets, res):
        print("order %d B=%2d %-28s median %.1f us = %.2f us/eval  (%s)  kernel %d" % (order, B, o, np.median(r), np.median(r) / B, " ".join("%.1f" % x for x in r), c.get_option("last_hess_kernel")), flush=True)
    ms.close()

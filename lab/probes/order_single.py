#!/usr/bin/env python3
"""One trajectory per launch (BASELINE config 3), residual + Jacobian at Pade orders 4 ... 10: HIP-event time per launch, option sets as
arguments (key=value,key=value), alternating in one process; every option set is checked bitwise against the first one at each order.
usage: order_single.py [orders=4,8,10] [optset ...]      e.g.  order_single.py 4,8,10 v4_flags=0 v4_flags=4"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

orders = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4,8,10").split(",")]
sets = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
system = synthetic.config_system(3)
t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
Zd = torch.from_numpy(t0.datavec.copy()[None]).cuda()
for order in orders:
    ctxs = []
    for o in sets:
        ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, 1, pade_order=order)
        ms.ctx.set_stream(stream.cuda_stream)
        for k, v in o.items():
            ms.ctx.set_option(k, int(v))
        ctxs.append(ms)
    c0 = ctxs[0].ctx
    dd = torch.empty(c0.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c0.jac_nnz, dtype=torch.float64, device="cuda")
    ref = None
    for ms, o in zip(ctxs, sets):
        vd.fill_(float("nan")); dd.fill_(float("nan"))
        ms.ctx.eval_jac_dev(Zd, dd, vd)
        ms.ctx.sync()
        assert torch.isfinite(vd).all() and torch.isfinite(dd).all(), (order, o)
        if ref is None:
            ref = (dd.clone(), vd.clone())
        else:
            assert torch.equal(dd, ref[0]) and torch.equal(vd, ref[1]), (order, o, "differs from the first option set")
    res = [[] for _ in sets]
    for rnd in range(6):
        idx = list(range(len(sets)))
        for i in (idx if rnd % 2 == 0 else idx[::-1]):
            c = ctxs[i].ctx
            for _ in range(10):
                c.eval_jac_dev(Zd, dd, vd)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(100):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            res[i].append(e0.elapsed_time(e1) / 100 * 1e3)
    for o, r, ms in zip(sets, res, ctxs):
        print("order %2d %-40s median %.2f us  (%s)  kernel %d" % (order, o, float(np.median(r)), " ".join("%.2f" % x for x in r), ms.ctx.get_option("last_kernel")), flush=True)
        ms.close()

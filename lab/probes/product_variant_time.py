#!/usr/bin/env python3
"""What the LDS operations of the generated product cost a one-trajectory launch: profile build (no stamps taken), timing variants of the product (WRONG results):
v4_variant 0 as shipped | 1 no ds_add_f64 | 2 no LDS operation in the epilogues.  HIP events, alternating.  usage: product_variant_time.py [orders=4,8,10]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
orders = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4,8,10").split(",")]
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    Zd = torch.from_numpy(t0.datavec.copy()[None]).cuda()
    for order in orders:
        ctxs = []
        for v in (0, 1, 2):
            ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, 1, pade_order=order)
            ms.ctx.set_stream(stream.cuda_stream)
            ms.ctx.set_option("v4_variant", v)
            ctxs.append(ms)
        c0 = ctxs[0].ctx
        dd = torch.empty(c0.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c0.jac_nnz, dtype=torch.float64, device="cuda")
        res = [[] for _ in ctxs]
        for rnd in range(6):
            idx = list(range(len(ctxs)))
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
        print("order %2d: as shipped %.2f us | no ds_add_f64 %.2f | no LDS operation in the products' epilogues %.2f" % (order, *[float(np.median(r)) for r in res]), flush=True)
        for ms in ctxs:
            ms.close()
finally:
    pa.build_library(force=True)

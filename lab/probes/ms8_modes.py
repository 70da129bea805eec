#!/usr/bin/env python3
"""8 (or B) trajectories per launch on `nbuf` separately allocated values arrays: the static split against the slice tickets (v4_ticket 0 / 1), alternating in one
process, HIP events; to be read beside the bare store patterns of stripe_probe / static_variants ON THE SAME BOX.  usage: ms8_modes.py [B=8] [nbuf=4] [key=value ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
opts = dict(kv.split("=") for kv in sys.argv[3:])
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
ctxs = {}
for name, tk in (("static", 0), ("tickets", 1)):
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4)
    ms.ctx.set_stream(stream.cuda_stream)
    ms.ctx.set_option("v4_ticket", tk)
    for k, v in opts.items():
        ms.ctx.set_option(k, int(v))
    ctxs[name] = ms
c0 = ctxs["static"].ctx
dd = torch.empty(c0.n_rows, dtype=torch.float64, device="cuda")
bufs = [torch.empty(c0.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
ref = None
for name, ms in ctxs.items():  # the same bits
    bufs[0].fill_(float("nan")); dd.fill_(float("nan"))
    ms.ctx.eval_jac_dev(Zd, dd, bufs[0]); ms.ctx.sync()
    if ref is None:
        ref = (dd.clone(), bufs[0].clone())
    else:
        print("%-8s bitwise equal to static: %s (non-finite: %d)" % (name, torch.equal(dd, ref[0]) and torch.equal(bufs[0], ref[1]), int((~torch.isfinite(bufs[0])).sum())), flush=True)
del ref
res = {n: [1e9] * nbuf for n in ctxs}
reps = 5 if B <= 16 else 2
for rnd in range(3):
    for name in (list(ctxs) if rnd % 2 == 0 else list(ctxs)[::-1]):
        c = ctxs[name].ctx
        for i, vd in enumerate(bufs):
            for _ in range(2):
                c.eval_jac_dev(Zd, dd, vd)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                c.eval_jac_dev(Zd, dd, vd)
            e1.record(stream)
            stream.synchronize()
            res[name][i] = min(res[name][i], e0.elapsed_time(e1) / reps * 1e3)
for name, r in res.items():
    print("%-8s us per launch of %d: min %.1f median %.1f max %.1f | %s" % (name, B, min(r), float(np.median(r)), max(r), " ".join("%.0f" % x for x in r)))

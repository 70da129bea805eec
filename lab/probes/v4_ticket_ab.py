#!/usr/bin/env python3
"""Kernel 4, launches of several trajectories: the static split (contiguous column ranges) against tickets, on several separately
allocated values arrays in ONE process (the time of the static split depends on where the array's pages live).  Checks that both
give the same bits, then times option sets alternating per buffer.
usage: v4_ticket_ab.py [trajectories=8] [buffers=8] [order=4] [key=value,key=value ...]   (each further argument = one option set)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nbuf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
order = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sets = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[4:]] or [{"v4_ticket": "0"}, {"v4_ticket": "1"}]
system = synthetic.config_system(3)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, device=0, pade_order=order)
c = ms.ctx
c.set_stream(stream.cuda_stream)
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
bufs = [torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
DEFAULTS = {"v4_ticket": 0, "v4_group": 0, "v4_ticket_ahead": 0, "v4_ticket_cols": 0, "v4_power_tiles": 0, "contiguous": -1, "v4_tail_mode": 3, "v4_flags": 0, "nt_stores": -1}


def apply(opts):
    for k, v in DEFAULTS.items():
        c.set_option(k, v)
    for k, v in opts.items():
        c.set_option(k, int(v))


def timeit(f, reps):
    for _ in range(2):
        f()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        f()
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# parity first: every option set against the first one, bitwise, delta and values
ref_d = ref_v = None
for o in sets:
    apply(o)
    dd.fill_(float("nan"))
    bufs[0].fill_(float("nan"))
    c.eval_jac_dev(Zd, dd, bufs[0])
    c.sync()
    d_, v_ = dd.clone(), bufs[0].clone()
    assert torch.isfinite(d_).all() and torch.isfinite(v_).all(), (o, "non-finite output")
    if ref_d is None:
        ref_d, ref_v = d_, v_
    else:
        assert torch.equal(d_, ref_d) and torch.equal(v_, ref_v), (o, "differs from the first option set", float((v_ - ref_v).abs().max()))
    print("parity ok", o, "last_kernel", c.get_option("last_kernel"), "ticket cols", c.get_option("last_v4_ticket"), flush=True)
del ref_d, ref_v
reps = 10 if B <= 16 else 4
res = {i: [] for i in range(len(sets))}
for bi, vd in enumerate(bufs):
    for i, o in enumerate(sets):
        apply(o)
        res[i].append(float(np.median([timeit(lambda: c.eval_jac_dev(Zd, dd, vd), reps) for _ in range(3)])))
print("%d trajectories per launch, order %d, %d buffers; us per launch (median of 3 x %d)" % (B, order, nbuf, reps))
for i, o in enumerate(sets):
    t = np.array(res[i])
    print("%-60s min %7.1f  median %7.1f  max %7.1f   %s" % (o, t.min(), np.median(t), t.max(), " ".join("%.0f" % x for x in t)))

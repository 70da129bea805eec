#!/usr/bin/env python3
"""Round 5: the decoupled design the round-4 review asked to be measured once -- compact producer (unique tiles + tails) and a pure
replicating expander (pcl_jac_expand_dev), serial on one stream and pipelined seed by seed on two streams -- against the fused launches
(slice tickets, static split, matrix-core kernel 3), alternating in one process on the same values array.  argv: seeds per launch (8)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = 20 if B <= 8 else 6
system = synthetic.config_system(3)
G0, Gj = system.G_drift, system.G_drives_array()
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
t0 = trajs[0]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.set_stream(sa)
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()


def ctx(batch, **opts):
    ms = pa.HipPadeMultistart(G0, Gj, t0, batch, pade_order=4)
    ms.ctx.set_stream(sa.cuda_stream)
    for k, v in opts.items():
        ms.ctx.set_option(k, v)
    return ms


fused = {"tickets": ctx(B), "static": ctx(B, v4_ticket=0), "k3": ctx(B, kernel_version=3)}
c = fused["tickets"].ctx
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
cv = torch.empty(c.compact_nnz, dtype=torch.float64, device="cuda")
per_full, per_comp, rows1 = c.jac_nnz // B, c.compact_nnz // B, c.n_rows // B
# one-seed contexts for the pipelined variant: producer s on stream A, expander s on stream B
prod = [ctx(1) for _ in range(B)]
expd = [ctx(1) for _ in range(B)]
for e in expd:
    e.ctx.set_stream(sb.cuda_stream)
evs = [torch.cuda.Event() for _ in range(B)]
ev_join = torch.cuda.Event()


def run_fused(name):
    fused[name].ctx.eval_jac_dev(Zd, dd, vd)


def run_compact():
    c.eval_jac_compact_dev(Zd, dd, cv)


def run_expand():
    c.jac_expand_dev(cv, vd)


def run_serial():
    c.eval_jac_compact_dev(Zd, dd, cv)
    c.jac_expand_dev(cv, vd)


def run_pipelined():
    for s in range(B):
        prod[s].ctx.eval_jac_compact_dev(Zd[s], dd[s * rows1:(s + 1) * rows1], cv[s * per_comp:(s + 1) * per_comp])
        evs[s].record(sa)
        sb.wait_event(evs[s])
        expd[s].ctx.jac_expand_dev(cv[s * per_comp:(s + 1) * per_comp], vd[s * per_full:(s + 1) * per_full])
    ev_join.record(sb)
    sa.wait_event(ev_join)


variants = {"fused_tickets": lambda: run_fused("tickets"), "fused_static": lambda: run_fused("static"), "fused_k3": lambda: run_fused("k3"),
            "compact_only": run_compact, "expand_only": run_expand, "compact_then_expand": run_serial, "pipelined_two_streams": run_pipelined}

# parity of the decoupled outputs with the fused launch (bitwise: one kernel family behind both)
run_fused("static")
sa.synchronize()
ref_v, ref_d = vd.clone(), dd.clone()
vd.zero_()
run_pipelined()
sa.synchronize()
sb.synchronize()
print("pipelined == fused (values, delta):", bool((vd == ref_v).all().item()), bool((dd == ref_d).all().item()), flush=True)
vd.zero_()
run_serial()
sa.synchronize()
print("serial == fused:", bool((vd == ref_v).all().item()), flush=True)

res = {k: [] for k in variants}
for rnd in range(5):
    names = list(variants) if rnd % 2 == 0 else list(variants)[::-1]
    for name in names:
        f = variants[name]
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sa.synchronize(); sb.synchronize()
        e0.record(sa)
        for _ in range(REPS):
            f()
        e1.record(sa)
        sa.synchronize(); sb.synchronize()
        res[name].append(e0.elapsed_time(e1) / REPS * 1e3)
bytes_full = vd.numel() * 8
for name, v in res.items():
    med = float(np.median(v))
    print("B=%d %-22s: %s  median %.1f us  (%.2f TB/s of the full values)" % (B, name, " ".join("%.1f" % x for x in v), med, bytes_full / med / 1e6), flush=True)

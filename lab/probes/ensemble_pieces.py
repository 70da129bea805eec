#!/usr/bin/env python3
"""Time the pieces of the ensemble step (config 4 share) separately: fused residual+Jacobian with per-member drift tiles,
objective, merit / shared-gradient payload."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

def timeit(fn, n=50, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

M, N = 8, 100
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
members = synthetic.config4_members(0, M)
traj = synthetic.synthetic_ensemble(members, N, seed=1)
Bs = pa.BilinearIntegrator(members, traj, pade_order=4)
c = Bs[0].ensemble.ctx
c.set_stream(stream.cuda_stream)
J = pa.UnitaryInfidelityObjective(np.eye(27, dtype=complex), [b.x_name for b in Bs], traj, Q=100.0, weights=np.full(M, 1 / M))
for nm in ("u", "du", "ddu"):
    J = J + pa.QuadraticRegularizer(nm, traj, 1e-2)
J.bind(Bs)
Zd = torch.from_numpy(traj.datavec).cuda()
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
ln, _ = c.merit_grad_len()
payload = torch.empty(ln + 1, dtype=torch.float64, device="cuda")
grad = torch.empty(c.z_len, dtype=torch.float64, device="cuda")
p0, p1 = payload[:1], payload[1:]
print("eval_jac (per-member drift) us:", timeit(lambda: c.eval_jac_dev(Zd, dd, vd)))
print("eval only us:", timeit(lambda: c.eval_dev(Zd, dd)))
print("objective us:", timeit(lambda: J.value_and_gradient_dev(Zd, p0, grad)))
print("merit_grad us:", timeit(lambda: c.merit_grad_dev(dd, None, vd, p1)))
mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
print("hess us:", timeit(lambda: c.hess_dev(Zd, mu, hv)))
# same shapes as independent trajectories (shared drift)
sysm = synthetic.config_system(3)
t0 = synthetic.synthetic_trajectory(sysm, N, seed=1)
ms = pa.HipPadeMultistart(sysm.G_drift, sysm.G_drives_array(), t0, M, pade_order=4)
c2 = ms.ctx; c2.set_stream(stream.cuda_stream)
Z2 = torch.from_numpy(np.stack([t0.datavec] * M)).cuda()
d2 = torch.empty(c2.n_rows, dtype=torch.float64, device="cuda")
v2 = torch.empty(c2.jac_nnz, dtype=torch.float64, device="cuda")
print("multistart eval_jac us:", timeit(lambda: c2.eval_jac_dev(Z2, d2, v2)))
print("multistart eval only us:", timeit(lambda: c2.eval_dev(Z2, d2)))
for g in (256, 384, 512, 768):
    c2.set_option("grid", g)
    print(" eval grid", g, timeit(lambda: c2.eval_dev(Z2, d2)))

# host-side (wall) cost per call, GPU idle-waiting excluded: enqueue 200 calls, then synchronise
import time
def wall(fn, n=200):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter() - t; torch.cuda.synchronize()
    return t1 / n * 1e6
c.set_option("grid", 0)
print("wall enqueue us: eval_jac", wall(lambda: c.eval_jac_dev(Zd, dd, vd)), "objective", wall(lambda: J.value_and_gradient_dev(Zd, p0, grad)),
      "merit", wall(lambda: c.merit_grad_dev(dd, None, vd, p1)), "eval", wall(lambda: c.eval_dev(Zd, dd)))
def step():
    c.eval_jac_dev(Zd, dd, vd)
    J.value_and_gradient_dev(Zd, payload[:1], grad)
    c.merit_grad_dev(dd, None, vd, payload[1:])
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): step()
    e1.record(); torch.cuda.synchronize()
    print("full step: wall us", (time.perf_counter() - t) / 100 * 1e6, "device us", e0.elapsed_time(e1) * 10)

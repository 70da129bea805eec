import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
system = synthetic.config_system(3)
def ens(tag):
    M, N = 8, 100
    members = synthetic.config4_members(0, M)
    traj = synthetic.synthetic_ensemble(members, N, seed=1)
    Bs = pa.BilinearIntegrator(members, traj, pade_order=4)
    c = Bs[0].ensemble.ctx; c.set_stream(stream.cuda_stream)
    J = pa.UnitaryInfidelityObjective(np.eye(27, dtype=complex), [b.x_name for b in Bs], traj, Q=100.0, weights=np.full(M, 1 / M))
    for nm in ("u", "du", "ddu"): J = J + pa.QuadraticRegularizer(nm, traj, 1e-2)
    J.bind(Bs)
    Zd = torch.from_numpy(traj.datavec).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"); vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    ln, _ = c.merit_grad_len(); payload = torch.empty(ln + 1, dtype=torch.float64, device="cuda"); grad = torch.empty(c.z_len, dtype=torch.float64, device="cuda")
    fns = {"eval_jac": lambda: c.eval_jac_dev(Zd, dd, vd), "objective": lambda: J.value_and_gradient_dev(Zd, payload[:1], grad), "merit": lambda: c.merit_grad_dev(dd, None, vd, payload[1:])}
    for name, fn in fns.items():
        for _ in range(5): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(100): fn()
        te = time.perf_counter() - t; torch.cuda.synchronize(); tt = time.perf_counter() - t
        print(tag, name, "enqueue us %.1f  total us %.1f" % (te * 1e4, tt * 1e4))
    for b in Bs: b.close()
ens("fresh")
t0 = synthetic.synthetic_trajectory(system, 100, seed=1)
for batch in (1, 8):
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, batch, pade_order=4)
    c = ms.ctx; c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack([t0.datavec] * batch)).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"); vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    for _ in range(50): c.eval_jac_dev(Zd, dd, vd)
    torch.cuda.synchronize(); ms.close(); del Zd, dd, vd
ens("after multistart")

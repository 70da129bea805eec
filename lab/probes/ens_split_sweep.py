"""Ensemble step (config 4 share, 8 members): stream / matrix workgroup split x fused or separate payload.  HIP-event time per step."""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

M, N = 8, 100
members = synthetic.config4_members(0, M)
traj = synthetic.synthetic_ensemble(members, N, seed=20260929 + 4)
Bs = pa.BilinearIntegrator(members, traj, device=0, pade_order=4)
c = Bs[0].ensemble.ctx
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
c.set_stream(st.cuda_stream)
Zd = torch.from_numpy(traj.datavec).cuda()
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
ln, _ = c.merit_grad_len()
out = torch.empty(ln, dtype=torch.float64, device="cuda")


def timeit(f, steps=60, warm=10):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


sws = [int(a) for a in sys.argv[1:]] or [-1, 96, 104, 112, 120, 128, 136]
for rep in range(2):
    for sw in sws:
        c.set_option("stream_workgroups", sw)
        t_plain = timeit(lambda: c.eval_jac_dev(Zd, dd, vd))
        t_sep = timeit(lambda: (c.eval_jac_dev(Zd, dd, vd), c.merit_grad_dev(dd, None, vd, out)))
        t_fus = timeit(lambda: c.eval_jac_merit_dev(Zd, None, dd, vd, out))
        print("stream_wg %4d (eff %3d): plain %.1f us | + separate payload %.1f | fused payload %.1f" % (sw, c.get_option("last_stream_workgroups"), t_plain, t_sep, t_fus), flush=True)

import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = 8
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4); c = ms.ctx
st = torch.cuda.Stream(); c.set_stream(st.cuda_stream)
with torch.cuda.stream(st):
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"); vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    cd = torch.empty(c.compact_nnz, dtype=torch.float64, device="cuda")
    c.eval_jac_compact_dev(Zd, dd, cd)
    for name, fn in (("compact", lambda: c.eval_jac_compact_dev(Zd, dd, cd)), ("expand", lambda: c.jac_expand_dev(cd, vd))):
        for _ in range(5): fn()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(50): fn()
        e1.record(st); st.synchronize()
        t = e0.elapsed_time(e1) / 50 * 1e3
        print(name, "us/launch %.1f  us/eval %.2f  GB/s %.0f" % (t, t / B, c.jac_nnz * 8 / t / 1e3))

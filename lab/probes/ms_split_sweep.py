"""Multistart share (config 5, B seeds per launch): stream / matrix workgroup split of fused kernel 3.  HIP-event time per launch."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4)
c = ms.ctx
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
c.set_stream(st.cuda_stream)
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")


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


sws = [int(a) for a in sys.argv[2:]] or [-1, 80, 88, 96, 104, 112, 120, 128]
for rep in range(2):
    for sw in sws:
        c.set_option("stream_workgroups", sw)
        t = timeit(lambda: c.eval_jac_dev(Zd, dd, vd))
        print("B %d stream_wg %4d (eff %3d): %.1f us per launch" % (B, sw, c.get_option("last_stream_workgroups"), t), flush=True)

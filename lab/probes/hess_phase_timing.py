#!/usr/bin/env python3
"""Per-phase s_memtime deltas of workgroup 0 / thread 0 of the Hessian kernel v2 (first items)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4)
c = ms.ctx
c.set_option("hess_kernel", 2); c.set_option("debug_timing", 1)
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda"); hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
c.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    c.hess_dev(Zd, mu, hv)
torch.cuda.synchronize()
out = (ctypes.c_int64 * 64)()
c._chk(c._L.pcl_debug_timing(c._h, out, 64))
t = np.array(out[:], dtype=np.int64)
t = t[t > 0]
names = ["prologue->item", "build G", "inputs", "P,E", "MFMA", "chunk out", "wave red", "barrier", "A2+out", "finish", "next item"]
print("stamps:", len(t), " deltas (s_memtime ticks, 100 MHz => x10 ns):")
print(np.diff(t).tolist())

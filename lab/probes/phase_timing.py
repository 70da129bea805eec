#!/usr/bin/env python3
"""Print the per-phase s_memtime deltas of workgroup 0 / matrix wave 0 (kernel v3, second item)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0: contiguous column ranges
ab = int(sys.argv[3]) if len(sys.argv) > 3 else 0
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4)
c = ms.ctx
c.set_option("kernel_version", 3); c.set_option("cols_per_slice", nc); c.set_option("debug_timing", 1); c.set_option("debug_ablate", ab)
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"); vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
c.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    c.eval_jac_dev(Zd, dd, vd)
torch.cuda.synchronize()
out = (ctypes.c_int64 * 64)()
c._chk(c._L.pcl_debug_timing(c._h, out, 64))
t = np.array(out[:], dtype=np.int64)
t = t[t > 0]
print("stamps:", len(t), " deltas (s_memtime ticks, 100 MHz => x10 ns):")
print(np.diff(t).tolist())

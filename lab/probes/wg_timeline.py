#!/usr/bin/env python3
"""Per-workgroup start / end times of kernel 3 (s_memrealtime, 100 MHz): which role finishes last in the role-split launch?"""
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
c.set_option("debug_timing", 1)
Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"); vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
c.set_stream(torch.cuda.current_stream().cuda_stream)
for ab in (0, 32, 8, 4):
    c.set_option("debug_ablate", ab)
    for _ in range(5):
        c.eval_jac_dev(Zd, dd, vd)
    torch.cuda.synchronize()
    W = 64 + 2 * 1024
    out = (ctypes.c_int64 * W)()
    c._chk(c._L.pcl_debug_timing(c._h, out, W))
    t = np.array(out[:], dtype=np.int64)
    ns = c.get_option("last_stream_workgroups")
    g = c.get_option("n_cu")
    st, en = t[64:64 + g], t[64 + 1024:64 + 1024 + g]
    t0 = st.min()
    us = lambda x: (x - t0) / 100.0
    print("ablate %d: stream WGs: start %.1f..%.1f us, end %.1f..%.1f (mean %.1f) | matrix WGs: start %.1f..%.1f, end %.1f..%.1f (mean %.1f)"
          % (ab, us(st[:ns]).min(), us(st[:ns]).max(), us(en[:ns]).min(), us(en[:ns]).max(), us(en[:ns]).mean(),
             us(st[ns:]).min(), us(st[ns:]).max(), us(en[ns:]).min(), us(en[ns:]).max(), us(en[ns:]).mean()), flush=True)

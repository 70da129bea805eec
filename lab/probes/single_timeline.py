#!/usr/bin/env python3
"""Single-trajectory launch of kernel 3 (profile build): prologue / first build / stream stamps of workgroup 0, and the
per-workgroup start / end spread (s_memrealtime, 100 MHz)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
pa.build_library(force=True, profile=True)
try:
    system = synthetic.config_system(3)
    t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, 1, pade_order=4)
    c = ms.ctx
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    Zd = torch.from_numpy(t0.datavec[None]).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"); vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    c.set_option("debug_timing", 1); c.set_option("profile_flags", 2)
    for _ in range(5): c.eval_jac_dev(Zd, dd, vd)
    torch.cuda.synchronize()
    W = 64 + 2 * 1024
    out = (ctypes.c_int64 * W)()
    c._chk(c._L.pcl_debug_timing(c._h, out, W))
    t = np.array(out[:], dtype=np.int64)
    s = t[40:51]
    print("WG0 stamps (cycles from entry): prologue-loads %d  barrier %d  ell %d  build0 %d  barrier %d | stream issued %d  drained %d | wave0 done %d | build0: controls %d  G(u) %d"
          % tuple(int(x - s[0]) for x in s[1:]))
    g = c.get_option("n_cu")
    st, en = t[64:64 + g], t[64 + 1024:64 + 1024 + g]
    ok = st > 0
    print("workgroups stamped:", int(ok.sum()), " start spread %.2f us, end: min %.2f mean %.2f max %.2f us after first start; duration mean %.2f us"
          % ((st[ok].max() - st[ok].min()) / 100.0, (en[ok].min() - st[ok].min()) / 100.0, (en[ok].mean() - st[ok].min()) / 100.0,
             (en[ok].max() - st[ok].min()) / 100.0, (en[ok] - st[ok]).mean() / 100.0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c.set_option("debug_timing", 0)
    e0.record()
    for _ in range(100): c.eval_jac_dev(Zd, dd, vd)
    e1.record(); torch.cuda.synchronize()
    print("launch-to-launch %.2f us" % (e0.elapsed_time(e1) * 10))
finally:
    pa.build_library(force=True)

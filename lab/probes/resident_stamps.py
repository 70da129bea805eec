#!/usr/bin/env python3
"""Where a resident evaluation's time goes: 100 MHz stamps per evaluation and workgroup (v4_flags & 2048).  usage: resident_stamps.py [order=4] [flags=0]"""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import _lib
_lib.build_library(lab=True)  # include/piccolo_hip_lab.h: the resident evaluator is not in the shipped library
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import resident_methods
resident_methods.attach()
from piccolo_jl_amd import synthetic

order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
opts = dict(kv.split("=") for kv in sys.argv[3:])
system = synthetic.config_system(3)
t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
Zd = torch.from_numpy(t0.datavec.copy()[None]).cuda()
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, 1, pade_order=order)
c = ms.ctx
c.set_option("v4_flags", flags | 2048)
for k, v in opts.items():
    c.set_option(k, int(v))
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
c.resident_start(Zd, dd, vd)
c.resident_post(200); c.resident_wait(10.0)   # warm
c.resident_stop()
c.resident_start(Zd, dd, vd)
import time
if os.environ.get("ROUND_TRIP"):
    tt = []
    for _ in range(16):
        t_ = time.perf_counter(); c.resident_post(1); c.resident_wait(10.0); tt.append((time.perf_counter() - t_) * 1e6)
    print("host: post + wait per request: " + " ".join("%.1f" % x for x in tt))
else:
    c.resident_post(16); c.resident_wait(10.0)
c.resident_stop()
out = np.zeros(16 * 256 * 4, dtype=np.int64)
c._chk(c._L.pcl_resident_stamps(c._h, out.ctypes.data, out.size))
s = out.reshape(16, 256, 4).astype(np.float64) / 100.0  # us
nwg = int((s[0, :, 0] > 0).sum())
s = s[:, :nwg, :]
base = s[0, :, 0].min()
s -= base
print("%d workgroups" % nwg)
np.set_printoptions(precision=1, suppress=True, linewidth=200)
print("evaluation: request seen (min / median / max over workgroups) | - | drained | arrival counted    [us since the first request was seen]")
for e in range(16):
    print("%2d: " % e + " | ".join("%7.1f %7.1f %7.1f" % (s[e, :, k].min(), np.median(s[e, :, k]), s[e, :, k].max()) for k in range(4)))
d = s[2:, :, :]
print("per workgroup, evaluations 2..15: seen -> drained %.1f us median (max %.1f), -> counted +%.1f, counted -> next seen %.1f" % (
    np.median(d[:, :, 2] - d[:, :, 0]), np.median((d[:, :, 2] - d[:, :, 0]).max(axis=1)), np.median(d[:, :, 3] - d[:, :, 2]), np.median(s[3:, :, 0] - s[2:-1, :, 3])))
print("evaluation period (max counted, e -> e+1): %s" % np.diff(s[:, :, 3].max(axis=1)))
dur = (s[2:15, :, 3] - s[2:15, :, 0]).mean(axis=0)
print("seen -> counted per workgroup, mean over evaluations 2..14, by workgroup mod 8: " + " ".join("%.1f" % dur[x::8].mean() for x in range(8)))
print("  by workgroup index, groups of 16: " + " ".join("%.0f" % dur[i:i + 16].mean() for i in range(0, nwg, 16)))
ms.close()

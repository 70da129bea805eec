import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import piccolo_jl_amd as pa
import pade_oracle as po
from piccolo_jl_amd import synthetic
so = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(so, 6, seed=900 + s) for s in range(3)]
ms = pa.HipPadeMultistart(so.G_drift, so.G_drives_array(), trajs[0], 3, pade_order=4)
c = ms.ctx
Zb = np.stack([t.datavec for t in trajs])
c.set_option("host_path", 1)
d_full, v_full = c.eval_jac(Zb)
d_full2, v_full2 = c.eval_jac(Zb)
print("path1 repeat equal", np.array_equal(d_full, d_full2), np.array_equal(v_full, v_full2))
c.set_option("host_path", 2)
for rep in range(3):
    d, v = c.eval_jac(Zb)
    print("path2 vs path1: d equal", np.array_equal(d, d_full), "v equal", np.array_equal(v, v_full), "max|dv|", np.abs(v - v_full).max(), "nan", np.isnan(v).sum(), "max|dd|", np.abs(d-d_full).max())
    bad = np.nonzero(v != v_full)[0]
    if len(bad): print("  first bad", bad[:5], "count", len(bad), "per", c.jac_per)

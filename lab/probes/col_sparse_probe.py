"""Pattern-compiled column kernel + stream-only kernel 3 (column_kernel = 2) against kernel 3's matrix role: agreement and launch times."""
import os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

system = synthetic.config_system(3)


def timeit(f, steps=40, warm=6):
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


for B in [int(a) for a in sys.argv[1:]] or [8, 1, 16]:
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4)
    c = ms.ctx
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd, vd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda"), torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    dd2, vd2 = torch.zeros_like(dd), torch.zeros_like(vd)
    cv, cv2 = torch.empty(c.compact_nnz, dtype=torch.float64, device="cuda"), torch.zeros(c.compact_nnz, dtype=torch.float64, device="cuda")
    c.set_option("contiguous", 1)
    c.eval_jac_dev(Zd, dd, vd)
    c.eval_jac_compact_dev(Zd, dd, cv)
    k_old = c.get_option("last_kernel")
    c.set_option("column_kernel", 2)
    c.eval_jac_dev(Zd, dd2, vd2)
    k_new = c.get_option("last_kernel")
    torch.cuda.synchronize()
    sc = float(vd.abs().max())
    print("B %d kernels %d -> %d | delta max|diff| %.2e (max %.2e) | jac max|diff| %.2e (max %.2e) equal blocks: %s" % (
        B, k_old, k_new, float((dd2 - dd).abs().max()), float(dd.abs().max()), float((vd2 - vd).abs().max()), sc, "n/a"), flush=True)
    dd2.zero_()
    c.eval_jac_compact_dev(Zd, dd2, cv2)
    torch.cuda.synchronize()
    print("   compact: kernel %d delta diff %.2e jac diff %.2e" % (c.get_option("last_kernel"), float((dd2 - dd).abs().max()), float((cv2 - cv).abs().max())), flush=True)
    for ck in (1, 2, 1, 2):
        c.set_option("column_kernel", ck)
        t_full = timeit(lambda: c.eval_jac_dev(Zd, dd2, vd2))
        t_comp = timeit(lambda: c.eval_jac_compact_dev(Zd, dd2, cv2))
        print("   column_kernel %d: full %.1f us per launch (%.2f us/eval) | compact %.1f us (%.2f us/eval)" % (ck, t_full, t_full / B, t_comp, t_comp / B), flush=True)
    ms.close()

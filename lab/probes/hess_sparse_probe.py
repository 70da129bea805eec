#!/usr/bin/env python3
"""Pattern-compiled Hessian kernel (hess_kernel=4) against kernel 3: agreement and launch times."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

system = synthetic.config_system(3)
def tm(c, Zd, mu, hv, reps=40):
    for _ in range(5): c.hess_dev(Zd, mu, hv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): c.hess_dev(Zd, mu, hv)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for batch in (1, 2, 8, 16):
    trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], batch, pade_order=4)
    c = ms.ctx
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
    h3 = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
    h4 = torch.full((c.hess_nnz,), float("nan"), dtype=torch.float64, device="cuda")
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    c.set_option("hess_kernel", 3)
    c.hess_dev(Zd, mu, h3); c.sync()
    t3 = tm(c, Zd, mu, h3)
    c.set_option("hess_kernel", 4)
    t0 = time.time()
    c.hess_dev(Zd, mu, h4); c.sync()
    jit_s = time.time() - t0
    err = (h3 - h4).abs().max().item(); ref = h3.abs().max().item()
    nbad = int(torch.isnan(h4).sum().item())
    t4 = tm(c, Zd, mu, h4)
    print("batch %2d: kernel3 %.1f us, kernel4 %.1f us (%.2f us/eval), first call %.2f s, max|diff| %.3e (max|H| %.3e), nan %d, last_hess_kernel %d"
          % (batch, t3, t4, t4 / batch, jit_s, err, ref, nbad, c.get_option("last_hess_kernel")), flush=True)
    if err > 1e-9 * max(ref, 1.0) or nbad:
        d = (h3 - h4).abs().cpu().numpy()
        per = c.hess_nnz // (batch * 99)
        blk = d.reshape(batch * 99, per)
        off = np.argmax(blk.max(0)); print("   worst offset in an interval's block:", off, "of", per, "| rows with error:", int((blk.max(1) > 1e-9).sum()))
        secs = [28, 28 + 6 * 1458, 28 + 7 * 1458, 28 + 13 * 1458, per]
        prev = 0
        for nm, e in zip(["scalars", "H3 (u,Xk)", "H4 (h,Xk)", "H5 (u,Xk+1)", "H6 (h,Xk+1)"], secs):
            print("   %-12s max diff %.3e" % (nm, blk[:, prev:e].max())); prev = e
    ms.close()

"""Times the Hessian-of-the-Lagrangian kernels (option hess_kernel 1 / 2) at BASELINE config 3, batch 8 and 1."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic


ABL = tuple(int(a) for a in sys.argv[1:]) or (0,)


def main():
    system = synthetic.config_system(3)
    G0, Gj = system.G_drift, system.G_drives_array()
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    for batch in (8, 1):
        trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(batch)]
        ms = pa.HipPadeMultistart(G0, Gj, trajs[0], batch, pade_order=4)
        c = ms.ctx
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
        for hk in (1, 2, 3):
            c.set_option("hess_kernel", hk)
            for grid in ((0,) if hk != 3 else (0, 128, 198, 256, 396)):
                c.set_option("grid", grid)
                for _ in range(5):
                    c.hess_dev(Zd, mu, hv)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    c.hess_dev(Zd, mu, hv)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 50 / batch
                print("batch %d hess_kernel %d grid %d: %.2f us/eval  (%.0f GB/s of output)" % (batch, hk, grid, us, c.hess_nnz / batch * 8 / us / 1e3), flush=True)
        c.set_option("grid", 0)
        ms.close()


if __name__ == "__main__":
    main()

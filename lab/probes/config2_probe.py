import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic
s2 = synthetic.config_system(2)
t2 = synthetic.synthetic_trajectory(s2, 100, seed=1)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for batch in (1, 8, 64):
        trajs = [synthetic.synthetic_trajectory(s2, 100, seed=i) for i in range(batch)]
        ms = pa.HipPadeMultistart(s2.G_drift, s2.G_drives_array(), t2, batch, pade_order=4)
        c = ms.ctx
        c.set_stream(stream.cuda_stream)
        Z = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        d2 = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        v2 = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        for kv in (0, 1, 2, 3):
            c.set_option("kernel_version", kv)
            for _ in range(20): c.eval_jac_dev(Z, d2, v2)
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(500): c.eval_jac_dev(Z, d2, v2)
            e1.record(stream); stream.synchronize()
            print("config 2 batch %d kernel_version %d: %.2f us/launch (last_kernel %d)" % (batch, kv, e0.elapsed_time(e1) / 500 * 1e3, c.get_option("last_kernel")), flush=True)
        ms.close()

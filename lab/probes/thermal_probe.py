#!/usr/bin/env python3
"""Is the slow / fast state of the write stream the HBM temperature?  Bursts of 25 launches after idle gaps of
0 .. 8 s, with the memory temperature (rocm-smi) read just before and after each burst."""
import os, re, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic


def temps():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"memory\) \(C\): ([0-9.]+)", out)
        j = re.search(r"junction\) \(C\): ([0-9.]+)", out)
        return (float(m.group(1)) if m else -1.0, float(j.group(1)) if j else -1.0)
    except Exception:
        return (-1.0, -1.0)


B = 8
system = synthetic.config_system(3)
trajs = [synthetic.synthetic_trajectory(system, 100, seed=1000 + i) for i in range(B)]
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], B, pade_order=4)
c = ms.ctx
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    c.set_stream(stream.cuda_stream)
    Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    for _ in range(3):
        c.eval_jac_dev(Zd, dd, vd)
    stream.synchronize()
    for gap in (8, 4, 2, 1, 0, 0, 0, 0, 8, 0, 0):
        time.sleep(gap)
        t_before = temps()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(25):
            c.eval_jac_dev(Zd, dd, vd)
        e1.record(stream)
        stream.synchronize()
        us = e0.elapsed_time(e1) / 25 / B * 1e3
        # keep the memory busy for a second after the last gaps = 0 (sustained load)
        t_after = temps()
        print("idle %ds -> burst of 25 launches: %.2f us/eval; memory temp before %.0f C, after %.0f C (junction %.0f -> %.0f)"
              % (gap, us, t_before[0], t_after[0], t_before[1], t_after[1]), flush=True)
        if gap == 0:
            for _ in range(3000):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
ms.close()

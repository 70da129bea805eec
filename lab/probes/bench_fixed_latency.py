"""Where the fixed ~45 us of a 20-step timed region go (bench.py time_steps): variants of the bracket around the same 20 launches."""
import os, sys, time, gc
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

system = synthetic.config_system(3)
t0_ = synthetic.synthetic_trajectory(system, 100, seed=1000)
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0_, 1, pade_order=4)
c = ms.ctx
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
c.set_stream(stream.cuda_stream)
Zd = torch.from_numpy(t0_.datavec[None]).cuda()
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
launch = lambda: c.eval_jac_dev(Zd, dd, vd)
L, h, pz, pd_, pv = c._L.pcl_eval_jac_dev, c._h, pa.integrators._ptr(Zd), pa.integrators._ptr(dd), pa.integrators._ptr(vd)
raw = lambda: L(h, pz, pd_, pv)


def run(variant, steps=20, warm=5, f=launch):
    gc.collect(); gc.disable()
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    if variant in ("events", "warm_events"):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if variant == "warm_events":  # the hipEvents exist before the timed region (torch creates them at the first record)
        ev0.record(); ev1.record()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    if variant in ("events", "warm_events"):
        ev0.record()
    for _ in range(steps):
        f()
    if variant in ("events", "warm_events"):
        ev1.record()
    if variant == "streamsync":
        stream.synchronize()
    torch.cuda.synchronize()
    w = time.perf_counter() - t0
    gc.enable()
    return w / steps * 1e6


for rep in range(3):
    print("events + device sync (bench.py): %.2f us/step | no events: %.2f | stream sync first: %.2f | raw ctypes call, no events: %.2f | 200 steps: %.2f | events created before the region: %.2f" % (
        run("events"), run("plain"), run("streamsync"), run("plain", f=raw), run("events", steps=200), run("warm_events")), flush=True)

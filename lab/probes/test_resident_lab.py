"""The resident evaluator's GPU tests (round 5), moved out of tests/ with the feature (include/piccolo_hip_lab.h).  Run on a GPU box with
    python -m pytest lab/probes/test_resident_lab.py -q
They build csrc/libpiccolo_hip_lab.so (-DPCL_LAB) first; the shipped library does not export pcl_resident_*."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import piccolo_jl_amd as pa
from piccolo_jl_amd import _lib

_lib.build_library(lab=True)
import resident_methods

resident_methods.attach()
from helpers import traj_from_Z
from oracle import pade_oracle as po
from piccolo_jl_amd import synthetic



@pytest.mark.parametrize("order", [4, 10])
def test_resident_evaluator_is_bitwise_the_launched_kernel(order):
    """pcl_resident_* (include/piccolo_hip.h): kernel 4's workgroups stay on the device and run one evaluation per posted request -- the same
    code compiled as a function, so residual and values are BITWISE those of pcl_eval_jac_dev [REF src/control/integrators.jl:620-640, 780-790:
    evaluate! + eval_jacobian, what a solver iteration calls], for the trajectory as it stands when the request is posted (rewritten in place
    between requests by a copy on another stream), for requests posted ahead, after the kernel has left on its idle limit (the next request
    starts it again) and after pcl_resident_stop (launches as before).  Also against the oracle at order 4."""
    import time

    import torch

    system = synthetic.config_system(3)
    t0 = synthetic.synthetic_trajectory(system, 12, seed=41)
    t1 = synthetic.synthetic_trajectory(system, 12, seed=42)
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, 1, pade_order=order)
    c = ms.ctx
    c.set_option("resident_idle_us", 2000)
    Z0 = torch.from_numpy(t0.datavec.copy()[None]).cuda()
    Z1 = torch.from_numpy(t1.datavec.copy()[None]).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    refs = []
    for Z in (Z0, Z1):
        c.eval_jac_dev(Z, dd, vd)
        c.sync()
        refs.append((dd.clone(), vd.clone()))
    if order == 4:
        so = po.config_system(3)
        lay = po.Layout.smooth_pulse(so.levels, len(so.G_drives), 12)
        dref, vref = ref_lib.eval_jac(t0.datavec.reshape(12, -1).copy(), lay, so.G_drift, np.array(so.G_drives))
        assert np.abs(refs[0][0].cpu().numpy() - dref.ravel()).max() <= 1e-12 and np.abs(refs[0][1].cpu().numpy() - vref.ravel()).max() <= 1e-12
    Zd = Z0.clone()
    dr = torch.full_like(dd, float("nan"))
    vr = torch.full_like(vd, float("nan"))
    assert c.resident_completed() == -1
    with pytest.raises(pa.PclError):
        c.resident_post(1)  # not started
    torch.cuda.synchronize()
    c.resident_start(Zd, dr, vr)
    with pytest.raises(pa.PclError):
        c.resident_start(Zd, dr, vr)  # already started
    c.resident_eval(10.0)
    assert c.resident_completed() == 1
    assert torch.equal(dr, refs[0][0]) and torch.equal(vr, refs[0][1])
    Zd.copy_(Z1)  # rewritten in place (a kernel of torch's stream)
    torch.cuda.synchronize()
    dr.fill_(float("nan"))
    torch.cuda.synchronize()
    c.resident_eval(10.0)
    assert torch.equal(dr, refs[1][0]) and torch.equal(vr, refs[1][1])
    c.resident_post(25)  # posted ahead
    c.resident_wait(10.0)
    assert c.resident_completed() == 27
    assert torch.equal(dr, refs[1][0]) and torch.equal(vr, refs[1][1])
    starts = c.get_option("resident_launches")
    time.sleep(0.05)  # 25 x the idle limit: the kernel has left by itself
    Zd.copy_(Z0)
    vr.fill_(float("nan"))
    torch.cuda.synchronize()
    c.resident_eval(10.0)
    assert c.get_option("resident_launches") > starts
    assert torch.equal(dr, refs[0][0]) and torch.equal(vr, refs[0][1])
    c.resident_stop()
    c.resident_stop()  # idempotent
    c.eval_jac_dev(Z1, dd, vd)  # launches as before
    c.sync()
    assert torch.equal(dd, refs[1][0]) and torch.equal(vd, refs[1][1])
    # a second session on other arrays
    dr2 = torch.full_like(dd, float("nan"))
    vr2 = torch.full_like(vd, float("nan"))
    torch.cuda.synchronize()
    c.resident_start(Z0, dr2, vr2)
    c.resident_eval(10.0)
    assert torch.equal(dr2, refs[0][0]) and torch.equal(vr2, refs[0][1])
    ms.close()  # (pcl_destroy stops a resident kernel)


def test_resident_evaluator_with_several_trajectories_per_request():
    """The resident kernel is whatever pcl_eval_jac_dev would launch with the static work split -- also for a context of several trajectories
    (contiguous column ranges, several intervals per workgroup): 5 trajectories of 40 knots per request, bitwise the launched values."""
    import torch

    system = synthetic.config_system(3)
    trajs = [synthetic.synthetic_trajectory(system, 40, seed=70 + i) for i in range(5)]
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), trajs[0], 5, pade_order=4)
    c = ms.ctx
    c.set_option("v4_ticket", 0)  # (the launched reference: the static split too -- the slice tickets give the same bits, asserted elsewhere)
    Z = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    c.eval_jac_dev(Z, dd, vd)
    c.sync()
    ref = (dd.clone(), vd.clone())
    dr = torch.full_like(dd, float("nan"))
    vr = torch.full_like(vd, float("nan"))
    torch.cuda.synchronize()
    c.resident_start(Z, dr, vr)
    c.resident_post(3)
    c.resident_wait(10.0)
    assert c.resident_completed() == 3
    c.resident_stop()
    assert torch.equal(dr, ref[0]) and torch.equal(vr, ref[1])
    ms.close()


def test_resident_evaluator_refuses_what_kernel_4_does_not_take():
    """Systems outside kernel 4 (here: d = 2, the small-system kernel's) get PCL_ESHAPE from pcl_resident_start, and nothing is launched."""
    import torch

    system = synthetic.config_system(1)
    t0 = synthetic.synthetic_trajectory(system, 10, seed=3)
    ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, 1, pade_order=4)
    c = ms.ctx
    Z = torch.from_numpy(t0.datavec.copy()[None]).cuda()
    dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
    vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
    with pytest.raises(pa.PclError) as ei:
        c.resident_start(Z, dd, vd)
    assert ei.value.code == -5
    c.eval_jac_dev(Z, dd, vd)  # the context still evaluates
    c.sync()
    assert torch.isfinite(vd).all()
    ms.close()

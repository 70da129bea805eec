#!/usr/bin/env python3
"""Resident evaluator (pcl_resident_*) against launches, one trajectory of BASELINE config 3: bitwise equality, the rate of requests posted
ahead, the round trip of one request (post + wait against launch + stream synchronise), the restart after the idle limit.
usage: resident_probe.py [order=4] [idle_us=5000] [key=value options ...]"""
import os, sys, time
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
idle = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
opts = dict(kv.split("=") for kv in sys.argv[3:])
system = synthetic.config_system(3)
t0 = synthetic.synthetic_trajectory(system, 100, seed=1000)
t1 = synthetic.synthetic_trajectory(system, 100, seed=1001)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
Z0 = torch.from_numpy(t0.datavec.copy()[None]).cuda()
Z1 = torch.from_numpy(t1.datavec.copy()[None]).cuda()
ms = pa.HipPadeMultistart(system.G_drift, system.G_drives_array(), t0, 1, pade_order=order)
c = ms.ctx
c.set_stream(stream.cuda_stream)
c.set_option("resident_idle_us", idle)
for k, v in opts.items():
    c.set_option(k, int(v))
Zd = Z0.clone()
dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
dr = torch.empty_like(dd)
vr = torch.empty_like(vd)
refs = []
for Z in (Z0, Z1):
    c.eval_jac_dev(Z, dd, vd)
    c.sync()
    refs.append((dd.clone(), vd.clone()))
stream.synchronize()


def wall(fn, n):
    t = time.perf_counter()
    fn(n)
    return (time.perf_counter() - t) / n * 1e6


# launches: queued back to back / one at a time
def launches_ahead(n):
    for _ in range(n):
        c.eval_jac_dev(Zd, dd, vd)
    c.sync()


def launches_round_trip(n):
    for _ in range(n):
        c.eval_jac_dev(Zd, dd, vd)
        c.sync()


for _ in range(3):
    la = wall(launches_ahead, 200)
    lr = wall(launches_round_trip, 200)
print("launches        : %.2f us per evaluation queued ahead, %.2f us per launch + synchronise" % (la, lr), flush=True)

dr.fill_(float("nan")); vr.fill_(float("nan"))
stream.synchronize()
c.resident_start(Zd, dr, vr)
c.resident_eval(5.0)
ok0 = torch.equal(dr, refs[0][0]) and torch.equal(vr, refs[0][1])
print("resident == launch (first trajectory): %s   completed %d" % (ok0, c.resident_completed()), flush=True)
# the trajectory rewritten in place by a copy on another stream
Zd.copy_(Z1)
stream.synchronize()
c.resident_eval(5.0)
ok1 = torch.equal(dr, refs[1][0]) and torch.equal(vr, refs[1][1])
print("resident == launch (trajectory rewritten in place): %s" % ok1, flush=True)


def res_ahead(n):
    c.resident_post(n)
    c.resident_wait(10.0)


def res_round_trip(n):
    for _ in range(n):
        c.resident_post(1)
        c.resident_wait(10.0)


for _ in range(3):
    ra = wall(res_ahead, 200)
    rr = wall(res_round_trip, 200)
print("resident        : %.2f us per evaluation posted ahead, %.2f us per post + wait   (starts so far: %d)" % (ra, rr, c.get_option("resident_launches")), flush=True)
ok2 = torch.equal(dr, refs[1][0]) and torch.equal(vr, refs[1][1])
time.sleep(max(0.05, 4 * idle * 1e-6))
Zd.copy_(Z0)
stream.synchronize()
c.resident_eval(5.0)
ok3 = torch.equal(dr, refs[0][0]) and torch.equal(vr, refs[0][1])
print("after the idle limit: equal %s, starts %d" % (ok3, c.get_option("resident_launches")), flush=True)
c.resident_stop()
# the context evaluates by launches as before
c.eval_jac_dev(Z1, dd, vd)
c.sync()
ok4 = torch.equal(dd, refs[1][0]) and torch.equal(vd, refs[1][1])
print("after stop: launches equal %s" % ok4, flush=True)
print("ALL EQUAL" if (ok0 and ok1 and ok2 and ok3 and ok4) else "MISMATCH")
ms.close()

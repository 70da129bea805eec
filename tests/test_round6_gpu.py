"""Round-6 GPU tests (``-m gpu``): bench.py starting its own ranks, graph capture of a multi-seed context, the RCCL entry points driven
from the C++ client of the C ABI, the general-order Hessian at the default order."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import piccolo_jl_amd as pa
from helpers import traj_from_Z
from oracle import pade_oracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 1 --force-dist ...` with NO launcher (RANK unset): bench.py re-runs itself under torch.distributed.run
    (self_launch), the rank initialises RCCL, and the parent's stdout carries exactly ONE JSON line with `rccl_ranks` 1 and
    `launcher` "self" -- the branch `python3 bench.py --gpus 8` takes on an 8-GPU node."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PCL_BENCH_SELF_LAUNCHED")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--no-extras", "--no-cpu-baseline"]
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["rccl_ranks"] == 1 and out["launcher"] == "self" and out["n_gpus"] == 1 and out["value"] > 0 and out["steps"] == 3


def test_bare_bench_refuses_more_gpus_than_the_box_has():
    """One GPU on this box: `--gpus 2` ends with status 2 and the reason on stderr, nothing on stdout (no half-started ranks)."""
    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert pr.returncode == 2 and pr.stdout.strip() == b"" and b"GPU(s) visible" in pr.stderr, (pr.returncode, pr.stderr[-500:])


def test_multi_seed_launch_is_graph_capturable_while_the_per_array_choice_is_open():
    """ADVICE (round 5): launches of several trajectories sample the static split against the slice tickets per values array with events on the
    launch stream (`v4_tune`).  A stream that is being captured must see none of that: no event record becomes a graph node, no query runs under
    capture; the captured launch takes the array's decided variant, else the tickets, and the sampling state does not move.  Capture on the
    SECOND call of a 4-seed config-3 context (the choice is still open), replay, compare bitwise with a direct launch."""
    import torch

    so = po.config_system(3)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Bn, N = 4, 100
    Zs = [po.synthetic_trajectory(so, N, seed=1000 + i)[0] for i in range(Bn)]
    lay = po.synthetic_trajectory(so, N, seed=1000)[1]
    ms = pa.HipPadeMultistart(G0, Gj, traj_from_Z(pa, Zs[0], lay), Bn, pade_order=4)
    c = ms.ctx
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(np.stack(Zs)).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        c.eval_jac_dev(Zd, dd, vd)  # first call: modules, attributes
        stream.synchronize()
        ref = (dd.clone(), vd.clone())
        assert c.get_option("last_v4_tune_choice") == -1  # still sampling
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            c.eval_jac_dev(Zd, dd, vd)
        assert c.get_option("last_v4_ticket") > 0  # undecided array under capture: the tickets
        for _ in range(3):
            dd.zero_()
            vd.zero_()
            g.replay()
            stream.synchronize()
            assert torch.equal(dd, ref[0]) and torch.equal(vd, ref[1])
        for _ in range(16):  # the sampling goes on afterwards, outside the capture, and still decides
            c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
        assert c.get_option("last_v4_tune_choice") in (0, 1) and torch.equal(vd, ref[1])
    ms.close()

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

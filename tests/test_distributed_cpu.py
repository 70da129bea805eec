"""world_size-2 gloo tests (CPU) of the N>1 path: units are sharded round-robin over ranks, each rank
evaluates only its own members, and ONE sum all_reduce of [merit | shared-control gradient] reproduces
the unsharded result.  The evaluator is replaced by the CPU oracle here (no GPU in this container);
the host logic under test is piccolo.jl_amd/distributed.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import piccolo_jl_amd as pa
from oracle import pade_oracle as po
from piccolo_jl_amd import distributed as pd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ensemble(M=5, N=7, seed=3):
    """Config-4-style ensemble: M perturbed drifts, shared controls, layout [U1..UM, dt, t, u]."""
    rng = np.random.default_rng(seed)
    base = po.config_system(1)
    d, m = base.levels, base.n_drives
    xd = 2 * d * d
    lay = po.Layout(d=d, m=m, N=N, z_dim=M * xd + 2 + m, x_off=0, u_off=M * xd + 2, dt_off=M * xd)
    Z = 0.3 * rng.standard_normal((N, lay.z_dim))
    Z[:, lay.dt_off] = 0.1 + 0.05 * rng.random(N)
    systems = [po.quantum_system((1 + 0.05 * (i - M // 2)) * base.H_drift, base.H_drives, [1.0, 1.0]) for i in range(M)]
    return systems, lay, Z, xd


def _member_outputs(systems, lay, Z, xd, members):
    dl, vl = [], []
    for i in members:
        s = systems[i]
        G0, Gj = s.G_drift, np.array(s.G_drives)
        dl.append(po.pade_residual(Z, lay, G0, Gj, 4, x_off=i * xd))
        vl.append(po.pade_jacobian_values(Z, lay, G0, Gj, 4, x_off=i * xd))
    return torch.from_numpy(np.stack(dl)).reshape(-1), torch.from_numpy(np.stack(vl)).reshape(-1)


def _worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    d_ = pd.init_process_group("gloo")
    systems, lay, Z, xd = _ensemble()
    M = len(systems)
    mine = pd.shard_indices(M, rank, world)
    delta, vals = _member_outputs(systems, lay, Z, xd, mine)
    w = torch.tensor([1.0 + 0.1 * i for i in mine], dtype=torch.float64)
    phi, gu, gdt = pd.constraint_merit_and_shared_gradient(delta, vals, len(mine), lay.K, lay.d, lay.m, w)
    # the objective part of the payload: every rank sums ITS members with the weights of the whole ensemble; the shared
    # regularisers enter on rank 0 only
    wall = np.array([1.0 + 0.1 * i for i in range(M)]) / M
    goal = po.PAULIS["X"]
    regs = [(lay.u_off, lay.m, 0.3, 2)] if rank == 0 else []
    Jr, _ = po.sampling_objective(Z, lay, [i * xd for i in mine], goal, wall[mine], 100.0, regs)
    payload = torch.cat([torch.tensor([Jr], dtype=torch.float64), phi.reshape(1), gu.reshape(-1), gdt.reshape(-1)])
    pd.reduce_payload(payload, d_)
    phi, gu, gdt = pd.reduce_merit_and_gradient(phi, gu, gdt, d_)
    per_unit = torch.tensor([float(i) for i in mine], dtype=torch.float64)
    allv = pd.gather_per_unit(per_unit, M, rank, world, d_)
    if rank == 0:
        torch.save(dict(phi=phi, gu=gu, gdt=gdt, allv=allv, payload=payload), out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_cover_exactly_once():
    for total, world in [(64, 8), (5, 2), (3, 4), (1, 1)]:
        seen = sorted(i for r in range(world) for i in pd.shard_indices(total, r, world))
        assert seen == list(range(total))
    assert pd.shard_indices(64, 3, 8) == list(range(3, 64, 8))  # 8 members per GPU at config 4
    with pytest.raises(ValueError):
        pd.shard_indices(4, 4, 4)


def test_jacobian_views_match_layout():
    d, m, K, B = 2, 2, 3, 2
    n, xd = 2 * d, 2 * d * d
    per = 2 * d * n * n + xd * (m + 1)
    vals = torch.arange(B * K * per, dtype=torch.float64)
    tail = pd.jacobian_views(vals, B, K, d, m)
    assert tail.shape == (B, K, d, m + 1, n)
    # (b=1, k=2, column c=1, drive l=1, row i=3) and the dt block (index m) of (b=0, k=1, c=0, i=2)
    assert tail[1, 2, 1, 1, 3].item() == (1 * K + 2) * per + 2 * d * n * n + (1 * (m + 1) + 1) * n + 3
    assert tail[0, 1, 0, m, 2].item() == 1 * per + 2 * d * n * n + (0 * (m + 1) + m) * n + 2


def test_world2_gloo_reduce_equals_unsharded(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    systems, lay, Z, xd = _ensemble()
    M = len(systems)
    delta, vals = _member_outputs(systems, lay, Z, xd, range(M))
    w = torch.tensor([1.0 + 0.1 * i for i in range(M)], dtype=torch.float64)
    phi, gu, gdt = pd.constraint_merit_and_shared_gradient(delta, vals, M, lay.K, lay.d, lay.m, w)
    assert torch.allclose(got["phi"], phi, rtol=1e-13, atol=0)
    assert torch.allclose(got["gu"], gu, rtol=1e-12, atol=1e-14)
    assert torch.allclose(got["gdt"], gdt, rtol=1e-12, atol=1e-14)
    assert torch.equal(got["allv"], torch.arange(M, dtype=torch.float64))
    # objective of the whole ensemble = sum of the ranks' shares (weights of the whole ensemble, regularisers once)
    wall = np.array([1.0 + 0.1 * i for i in range(M)]) / M
    J_all, _ = po.sampling_objective(Z, lay, [i * xd for i in range(M)], po.PAULIS["X"], wall, 100.0, [(lay.u_off, lay.m, 0.3, 2)])
    assert abs(float(got["payload"][0]) - J_all) < 1e-12 * max(1.0, abs(J_all))
    assert got["payload"].numel() == 2 + lay.K * lay.m + lay.K
    assert torch.allclose(got["payload"][1], phi, rtol=1e-13, atol=0) and torch.allclose(got["payload"][2 : 2 + lay.K * lay.m].view(lay.K, lay.m), gu, rtol=1e-12, atol=1e-14)
    # the reduced gradient is the gradient of the merit function w.r.t. the shared controls (finite differences)
    def merit(Zp):
        d_, _ = _member_outputs(systems, lay, Zp, xd, range(M))
        return float(0.5 * (w[:, None] * d_.view(M, -1) ** 2).sum())
    eps = 1e-6
    for (k, l) in [(0, 0), (2, 1), (lay.K - 1, 0)]:
        Zp, Zm = Z.copy(), Z.copy()
        Zp[k, lay.u_off + l] += eps
        Zm[k, lay.u_off + l] -= eps
        assert abs((merit(Zp) - merit(Zm)) / (2 * eps) - float(gu[k, l])) < 1e-6
    Zp, Zm = Z.copy(), Z.copy()
    Zp[1, lay.dt_off] += eps
    Zm[1, lay.dt_off] -= eps
    assert abs((merit(Zp) - merit(Zm)) / (2 * eps) - float(gdt[1])) < 1e-6


def test_single_process_reduce_is_identity():
    phi, gu, gdt = torch.tensor(2.0, dtype=torch.float64), torch.ones(3, 2, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)
    a, b, c = pd.reduce_merit_and_gradient(phi, gu, gdt, None)
    assert a.item() == 2.0 and torch.equal(b, gu) and torch.equal(c, gdt)


def test_every_unit_has_exactly_one_rank():
    """BASELINE configs 4 / 5: 64 members / seeds over 1, 2, 4, 8 ranks -- unit b on rank b mod world, the
    ranks' shares are disjoint, equal in size and cover 0..63; bench.py shards with the same function."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world in (1, 2, 4, 8):
        shares = [pd.shard_indices(64, r, world) for r in range(world)]
        assert sorted(u for s in shares for u in s) == list(range(64))
        assert all(len(s) == 64 // world for s in shares)
        assert all(u % world == r for r, s in enumerate(shares) for u in s)
        assert [bench.units_of_rank(64, r, world) for r in range(world)] == shares
    with pytest.raises(ValueError):
        pd.shard_indices(64, 8, 8)
    # the members a rank builds are those of its share (per-member perturbations are seeded by the member's index)
    from piccolo_jl_amd import synthetic

    a = synthetic.config4_members(0, 0, indices=[1, 9])
    b = synthetic.config4_members(1, 1) + synthetic.config4_members(9, 1)
    assert all(np.array_equal(x.G_drift, y.G_drift) for x, y in zip(a, b))


def test_bench_without_launcher_refuses_when_the_devices_are_missing():
    """`python bench.py --gpus 2` started bare takes the self-launch branch (bench.self_launch); on a box without two GPUs it must stop
    with status 2 and the reason -- not spawn ranks that die one by one."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the refusal branch is not reachable")
    pr = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert pr.returncode == 2 and pr.stdout.strip() == b"" and b"GPU(s) visible" in pr.stderr

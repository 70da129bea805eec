"""One process per GPU: sharding of independent units (ensemble members / multistart seeds) over
ranks and the one real exchange of the path -- a sum-reduce of the merit scalar and of the
gradient with respect to the SHARED controls and timesteps.

The evaluator itself needs no collective: Jacobian rows are member-private and never leave the
GPU that produced them.  `torch.distributed` is plumbing here: backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU (tests).  Payload of the reduce: (1 + m*K + K) doubles (~5.6 KB at config 4):
latency-bound, one all_reduce per evaluation.

Reference context: members of a SamplingTrajectory share `u`, `dt` and own a state copy each
[REF src/quantum/trajectories/sampling_trajectory.jl:207-237]; the reference sums the per-member
objectives inside one NLP [REF src/control/templates/sampling_problem.jl:381-387] and has no
multi-process path at all.
"""
import os

import torch


def shard_indices(total, rank, world):
    """Round-robin ownership: unit b lives on rank b mod world (8 members per GPU at config 4)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, total, world))


def init_process_group(backend=None):
    """Initialise torch.distributed from the launcher's environment (RANK, WORLD_SIZE, MASTER_*)."""
    import torch.distributed as dist

    if dist.is_initialized():
        return dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    dist.init_process_group(backend)
    return dist


def jacobian_views(vals, batch, K, d, m, cols=None):
    """View into the Jacobian value buffer (order: include/piccolo_hip.h): the tail as
    ``[batch, K, cols, m+1, n]`` (state column, drive l or dt at index m, row) without copying."""
    C = d if cols is None else cols
    n = 2 * d
    per = 2 * C * n * n + n * C * (m + 1)
    v = vals.view(batch, K, per)
    return v[:, :, 2 * C * n * n :].reshape(batch, K, C, m + 1, n)


def constraint_merit_and_shared_gradient(delta, vals, batch, K, d, m, weights=None, cols=None):
    """phi = sum_i w_i/2 |delta_i|^2 over this rank's members and its gradient with respect to the
    shared variables: g_u[k, l] = sum_i w_i <d delta_ik / d u_l, delta_ik>, g_dt[k] likewise.
    Works on any device (torch ops on views of the evaluator's output buffers)."""
    C = d if cols is None else cols
    n = 2 * d
    dl = delta.view(batch, K, C, n)
    w = torch.ones(batch, dtype=delta.dtype, device=delta.device) if weights is None else weights.to(delta)
    tail = jacobian_views(vals, batch, K, d, m, cols)
    phi = 0.5 * torch.einsum("b,bkci,bkci->", w, dl, dl)
    g = torch.einsum("b,bkcli,bkci->kl", w, tail, dl)  # [K, m+1]
    return phi, g[:, :m].contiguous(), g[:, m].contiguous()


def reduce_merit_and_gradient(phi, g_u, g_dt, dist=None):
    """One sum all_reduce of [phi | g_u | g_dt] over the ranks; returns the reduced pieces."""
    buf = torch.cat([phi.reshape(1), g_u.reshape(-1), g_dt.reshape(-1)])
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        _all_reduce_sum(buf, dist)
    K, m = g_u.shape
    return buf[0], buf[1 : 1 + K * m].view(K, m), buf[1 + K * m :]


def _all_reduce_sum(t, dist):
    """Sum all-reduce in place.  RCCL ("nccl") takes device tensors; a gloo group (CPU tests, or several ranks sharing one GPU) is
    handed a host copy."""
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def reduce_payload(payload, dist=None):
    """The all-reduce of a sharded ensemble step: ``payload`` = ``[objective | merit | J^T lam on u | on dt]`` as filled on
    the device by ``pcl_objective_dev`` + ``pcl_merit_grad_dev`` (each rank: its own members, weights w_i of the WHOLE
    ensemble; the shared regularisers are bound on rank 0 only, or with R / world everywhere).  In place; one sum.
    Issued whenever a process group exists -- also over ONE rank, so that a single-GPU run under the launcher executes the
    collective's whole code path."""
    if dist is not None and dist.is_initialized():
        _all_reduce_sum(payload, dist)
    return payload


def gather_per_unit(values, total, rank, world, dist=None):
    """Multistart (config 5): collect one scalar per seed from every rank into seed order."""
    out = torch.zeros(total, dtype=values.dtype, device=values.device)
    idx = shard_indices(total, rank, world)
    out[idx] = values
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        _all_reduce_sum(out, dist)
    return out

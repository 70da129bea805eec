#!/usr/bin/env python3
"""bench.py -- constraint+Jacobian evals/sec of the 3-transmon (d=27) unitary problem, N=100 knots.

One *eval* = one fused computation of the full dynamics residual delta (x_dim*(N-1) doubles) and all
Jacobian values in final triplet order for all N-1 intervals of ONE trajectory.
One *step* = one launch of the fused kernel over this rank's batch of independent multistart seeds
(BASELINE.json config 5's share: 64 seeds / 8 GPUs = 8 seeds per GPU; weak scaling, no data-path
collective -- seeds are independent NLPs).  Inputs and outputs are resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- algorithmic HBM bytes per launch / measured kernel time vs the 8 TB/s HBM3E peak
  cpu_baseline -- the oracle's C restatement (oracle/pade_ref.c, OpenMP) timed on this host (rank 0, N=1)
and `single_trajectory`: the same metric at batch 1 (BASELINE config 3 strictly: one NLP, one launch).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec); ~6.3 TB/s achievable


def algorithmic_bytes_per_eval(d, m, N, z_dim):
    """read z_k once per interval + write delta + write all Jacobian values."""
    n, xd, K = 2 * d, 2 * d * d, N - 1
    per_interval = z_dim * 8 + xd * 8 + (2 * d * n * n + xd * (m + 1)) * 8
    return per_interval * K


def time_steps(launch, steps, warmup, torch, dist):
    for _ in range(warmup):
        launch()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        launch()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8, help="multistart seeds per GPU (64 seeds / 8 GPUs)")
    ap.add_argument("--knots", type=int, default=100)
    ap.add_argument("--cols-per-slice", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the batch-1 (single trajectory) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl) even for one rank: exercises the multi-rank code path on one GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip the Hessian / compact / host-delivered / config-2 rates")
    args = ap.parse_args()

    import torch

    import piccolo_jl_amd as pa
    from piccolo_jl_amd import synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the evaluator has no CPU path)")
    torch.cuda.set_device(local)
    dist = None
    saved_stdout = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout when the communicator comes up; the contract is ONE JSON line on stdout,
        # so stdout (fd 1) points at stderr until the line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    stream = torch.cuda.Stream()  # the launch stream; HIP events below are recorded on it
    torch.cuda.set_stream(stream)
    system = synthetic.config_system(3)
    N, B = args.knots, args.batch
    d, m = system.levels, system.n_drives
    # multistart seeds s = 0..: default_rng(1000 + s)  (SURVEY 8(d)); this rank owns seeds rank*B .. rank*B+B-1
    trajs = [synthetic.synthetic_trajectory(system, N, seed=1000 + rank * B + i) for i in range(B)]
    t0 = trajs[0]
    G0, Gj = system.G_drift, system.G_drives_array()

    def run_case(batch, steps, warmup):
        ms = pa.HipPadeMultistart(G0, Gj, t0, batch, device=local)
        c = ms.ctx
        if args.cols_per_slice:
            c.set_option("cols_per_slice", args.cols_per_slice)
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs[:batch]])).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        wall, dev = time_steps(lambda: c.eval_jac_dev(Zd, dd, vd), steps, warmup, torch, dist)
        chk = float(dd.abs().max().item())
        assert np.isfinite(chk) and chk > 0
        nc = c.get_option("effective_cols_per_slice")
        lk = c.get_option("last_kernel")
        ns = c.get_option("last_stream_workgroups")
        ms.close()
        del Zd, dd, vd
        return wall, dev, nc, lk, ns

    wall, dev, nc, lk, ns = run_case(B, args.steps, args.warmup)
    t = torch.tensor([wall, dev], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, dev = float(t[0]), float(t[1])
    evals = world * B * args.steps
    abytes = algorithmic_bytes_per_eval(d, m, N, t0.dim)

    out = {
        "metric": "constraint+Jacobian evals/sec, 3-transmon d=27 unitary, T=100 knots",
        "value": evals / wall,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE config 3 problem (MultiTransmonSystem 3x3 levels, d=27, x_dim=1458, m=6, z_dim=%d, N=%d knots), "
            "Pade-4 fused residual+Jacobian (16,599,330 nnz/eval), %d multistart seeds per GPU in one launch (config 5 share), "
            "outputs left in HBM" % (t0.dim, N, B),
            "seeds_per_gpu": B,
            "total_seeds": B * world,
            "cols_per_slice": nc,
            "stream_workgroups": ns,
            "parallelism": "seeds sharded over %d rank(s), no data-path collective" % world,
        },
    }
    kernel_s = dev / args.steps  # HIP events on the launch stream around the K back-to-back launches
    out["roofline"] = {
        "bound": "hbm",
        "achieved": abytes * B / kernel_s / 1e9,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": abytes * B / kernel_s / 1e9 / HBM_PEAK_GBS,
        "traffic": None,
        "kernel": {31: "pcl_fused_kernel_v3<2,27,6,2>", 30: "pcl_fused_kernel_v3<2,0,0,0>", 41: "pcl_fused_kernel_v4<1,27,6,3>", 40: "pcl_fused_kernel_v4<1,0,0,0>", 21: "pcl_fused_kernel_v2<true,1,27,6,3>",
                   20: "pcl_fused_kernel_v2<true,1,0,0,0>"}.get(lk, "pcl_fused_kernel (id %d)" % lk)
        + ((" (persistent; 1 workgroup/CU; contiguous column ranges; %d workgroups stream the B+- blocks, the others do the column work "
            "with 8 matrix waves)" % ns) if lk // 10 == 3 and ns > 0 else
           " (persistent; 1 workgroup/CU; 4 MFMA waves + 4 store-stream waves; one barrier per item)" if lk // 10 == 3 else
           " (persistent; 2 workgroups/CU; 4 MFMA waves + 4 store-stream waves each)"),
        "kernel_us": kernel_s * 1e6,
        "algorithmic_bytes_per_launch": abytes * B,
    }
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):  # HBM bytes per launch from the rocprofv3 --pmc passes (scripts/profile.sh), same batch only
        try:
            tr = json.load(open(pmc))
            if tr.get("batch") == B and tr.get("knots") == N:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_single:
        w1, d1, nc1, lk1, ns1 = run_case(1, max(args.steps, 200), args.warmup)
        st = max(args.steps, 200)
        out["single_trajectory"] = {
            "evals_per_s": st / w1,
            "us_per_eval_wall": w1 / st * 1e6,
            "us_per_eval_kernel": d1 / st * 1e6,
            "cols_per_slice": nc1,
            "kernel_id": lk1,
            "hbm_GBps": abytes / (d1 / st) / 1e9,
        }
    if rank == 0 and world == 1 and not args.no_extras:
        # the other rates of the same path, reported beside the headline (never as `value`)
        ex = {}
        st = max(20, min(args.steps, 100))
        ms = pa.HipPadeMultistart(G0, Gj, t0, B, device=local)
        c = ms.ctx
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs])).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        mu = torch.randn(c.n_rows, dtype=torch.float64, device="cuda")
        hv = torch.empty(c.hess_nnz, dtype=torch.float64, device="cuda")
        w, dv = time_steps(lambda: c.hess_dev(Zd, mu, hv), st, 5, torch, None)
        ex["hessian_of_lagrangian"] = {"evals_per_s": B * st / w, "us_per_eval_kernel": dv / st / B * 1e6, "batch": B,
                                       "nnz_per_eval": c.hess_nnz // B}
        cv = torch.empty(c.compact_nnz, dtype=torch.float64, device="cuda")
        w, dv = time_steps(lambda: c.eval_jac_compact_dev(Zd, dd, cv), st, 5, torch, None)
        ex["compact_jacobian"] = {"evals_per_s": B * st / w, "us_per_eval_kernel": dv / st / B * 1e6, "batch": B,
                                  "values_per_eval": c.compact_nnz // B}
        w, dv = time_steps(lambda: c.eval_dev(Zd, dd), st, 5, torch, None)
        ex["residual_only"] = {"evals_per_s": B * st / w, "us_per_eval_kernel": dv / st / B * 1e6, "batch": B}
        ms.close()
        del Zd, dd, mu, hv, cv
        # host-delivered: the host-pointer entry point (H2D of Z, kernel, D2H of delta + 132.8 MB of values), one trajectory
        it = pa.HipPadeIntegrator(G0, Gj, t0, device=local)
        hd = np.empty(it.ctx.n_rows)
        hvals = np.empty(it.ctx.jac_nnz)
        it.ctx.eval_jac(t0.datavec, hd, hvals)
        th = time.perf_counter()
        for _ in range(5):
            it.ctx.eval_jac(t0.datavec, hd, hvals)
        th = (time.perf_counter() - th) / 5
        ex["host_delivered"] = {"evals_per_s": 1.0 / th, "ms_per_eval": th * 1e3, "GBps_over_pcie": hvals.nbytes / th / 1e9,
                                "note": "pageable numpy buffers"}
        it.close()
        # BASELINE config 2 (CNOT, d=4, N=100): launch-bound, report us/eval
        s2 = synthetic.config_system(2)
        t2 = synthetic.synthetic_trajectory(s2, 100, seed=20260929 + 2)
        i2 = pa.HipPadeIntegrator(s2.G_drift, s2.G_drives_array(), t2, device=local)
        i2.ctx.set_stream(stream.cuda_stream)
        Z2 = torch.from_numpy(t2.datavec).cuda()
        d2 = torch.empty(i2.ctx.n_rows, dtype=torch.float64, device="cuda")
        v2 = torch.empty(i2.ctx.jac_nnz, dtype=torch.float64, device="cuda")
        w, dv = time_steps(lambda: i2.ctx.eval_jac_dev(Z2, d2, v2), 200, 20, torch, None)
        ex["config2_cnot"] = {"evals_per_s": 200 / w, "us_per_eval_wall": w / 200 * 1e6, "us_per_eval_kernel": dv / 200 * 1e6}
        i2.close()
        out["other_rates"] = ex
    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            from oracle import pade_oracle as po
            from oracle import ref_lib

            so = po.config_system(3)
            lay = po.Layout.smooth_pulse(d, m, N)
            Z = trajs[0].datavec.reshape(N, t0.dim)
            avail = os.cpu_count() or 1
            try:
                avail = len(os.sched_getaffinity(0))
            except Exception:
                pass
            G0o, Gjo = so.G_drift, np.array(so.G_drives)
            outbuf = (np.empty((lay.K, lay.x_dim)), np.empty((lay.K, po.jac_nnz_per_interval(lay))))
            ref_lib.eval_jac(Z, lay, G0o, Gjo, nthreads=min(avail, lay.K), out=outbuf)  # warm-up (page faults)
            # the port parallelises over the K=99 intervals and is memory-write-bound: pick the best thread count
            best_t, cores = 1e9, 1
            for nt in sorted({1, 8, 16, 32, 64, min(avail, lay.K)}):
                if nt > avail:
                    continue
                t1 = time.perf_counter()
                for _ in range(2):
                    ref_lib.eval_jac(Z, lay, G0o, Gjo, nthreads=nt, out=outbuf)
                t1 = (time.perf_counter() - t1) / 2
                if t1 < best_t:
                    best_t, cores = t1, nt
            n, tc = 0, time.perf_counter()
            while time.perf_counter() - tc < args.cpu_seconds:
                ref_lib.eval_jac(Z, lay, G0o, Gjo, nthreads=cores, out=outbuf)
                n += 1
            el = time.perf_counter() - tc
            out["cpu_baseline"] = {
                "value": n / el,
                "unit": "evals/s",
                "cores": cores,
                "kind": "port",
                "sample": "%d evals of one config-3 trajectory (N=%d) in %.1f s, oracle/pade_ref.c (analytic Pade-4, OpenMP over "
                "intervals, gcc -O3 -march=x86-64-v3), outputs preallocated, best of thread counts up to %d available cores"
                % (n, N, el, avail),
            }
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Sweep kernel options on the GPU box (kernel time from HIP events on the launch stream)."""
import argparse, itertools, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import piccolo_jl_amd as pa
from piccolo_jl_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="+", default=[1, 8])
ap.add_argument("--nc", type=int, nargs="+", default=[0])
ap.add_argument("--nt", type=int, nargs="+", default=[0])
ap.add_argument("--mfma", type=int, nargs="+", default=[1])
ap.add_argument("--ablate", type=int, nargs="+", default=[0])
ap.add_argument("--kernel", type=int, nargs="+", default=[0])
ap.add_argument("--grid", type=int, nargs="+", default=[0])
ap.add_argument("--specialize", type=int, nargs="+", default=[1])
ap.add_argument("--cpp", type=int, nargs="+", default=[6])
ap.add_argument("--contig", type=int, nargs="+", default=[-1])
ap.add_argument("--nstream", type=int, nargs="+", default=[-1])
ap.add_argument("--flat", type=int, nargs="+", default=[0])
ap.add_argument("--snc", type=int, nargs="+", default=[0])
ap.add_argument("--sdyn", type=int, nargs="+", default=[1])
ap.add_argument("--sxcd", type=int, nargs="+", default=[0])
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--knots", type=int, default=100)
args = ap.parse_args()
system = synthetic.config_system(3)
N = args.knots
trajs = [synthetic.synthetic_trajectory(system, N, seed=1000 + i) for i in range(max(args.batch))]
G0, Gj = system.G_drift, system.G_drives_array()
d, m, zd = system.levels, system.n_drives, trajs[0].dim
abytes = (zd * 8 + 2 * d * d * 8 + (2 * d * 4 * d * d + 2 * d * d * (m + 1)) * 8) * (N - 1)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for B in args.batch:
        ms = pa.HipPadeMultistart(G0, Gj, trajs[0], B, pade_order=4)
        c = ms.ctx
        c.set_stream(stream.cuda_stream)
        Zd = torch.from_numpy(np.stack([t.datavec for t in trajs[:B]])).cuda()
        dd = torch.empty(c.n_rows, dtype=torch.float64, device="cuda")
        vd = torch.empty(c.jac_nnz, dtype=torch.float64, device="cuda")
        print("occupancy_v2 (blocks*1e6 + lds bytes):", c.get_option("occupancy_v2"), flush=True)
        for kv, gr, sp, cp, nc, nt, mf, ab, cg, ns, fl, sn, sy, sx in itertools.product(args.kernel, args.grid, args.specialize, args.cpp, args.nc, args.nt, args.mfma, args.ablate, args.contig, args.nstream, args.flat, args.snc, args.sdyn, args.sxcd):
            c.set_option("kernel_version", kv); c.set_option("grid", gr); c.set_option("specialize", sp); c.set_option("copies_per_piece", cp)
            c.set_option("cols_per_slice", nc); c.set_option("contiguous", cg); c.set_option("stream_workgroups", ns); c.set_option("aligned_stream", fl); c.set_option("stream_piece_cols", sn); c.set_option("stream_dynamic", sy); c.set_option("stream_xcds", sx); c.set_option("nt_stores", nt); c.set_option("use_mfma", mf); c.set_option("debug_ablate", ab)
            for _ in range(10):
                c.eval_jac_dev(Zd, dd, vd)
            stream.synchronize()
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(args.steps):
                    c.eval_jac_dev(Zd, dd, vd)
                e1.record(stream)
                stream.synchronize()
                best = min(best, e0.elapsed_time(e1) / args.steps * 1e3)
            print(json.dumps(dict(batch=B, sxcd=sx, sdyn=sy, snc=sn, flat=fl, contig=cg, nstream=ns, kernel=kv, grid=gr, spec=sp, cpp=cp, nc=nc, eff_nc=c.get_option("effective_cols_per_slice"), nt=nt, mfma=mf, ablate=ab,
                                  us_per_launch=round(best, 2), us_per_eval=round(best / B, 2), GBps=round(abytes * B / best / 1e3, 1))), flush=True)
        ms.close()

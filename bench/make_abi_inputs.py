#!/usr/bin/env python3
"""Write the inputs of bench/bench_abi.cpp: BASELINE config 3 (MultiTransmonSystem 3 x 3 levels, N = 100 knots) as one
little-endian blob: int32 header [d, m, N, z_dim, x_off, u_off, dt_off, 0] | G0 (n*n, column-major) | Gj (m*n*n) | Z (N*z_dim).
Host-side only (numpy): needs no GPU and no oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piccolo_jl_amd import synthetic

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "config3_inputs.bin")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
system = synthetic.config_system(3)
t = synthetic.synthetic_trajectory(system, N, seed=1000)
d, m = system.levels, system.n_drives
comp = t.components
hdr = np.array([d, m, N, t.dim, comp["Ũ⃗"].start, comp["u"].start, comp["Δt"].start, 0], dtype="<i4")
G0 = np.asfortranarray(system.G_drift).ravel(order="F")
Gj = np.concatenate([np.asfortranarray(g).ravel(order="F") for g in system.G_drives_array()])
with open(out, "wb") as f:
    f.write(hdr.tobytes())
    f.write(G0.astype("<f8").tobytes())
    f.write(Gj.astype("<f8").tobytes())
    f.write(np.ascontiguousarray(t.datavec, dtype="<f8").tobytes())
print(out, os.path.getsize(out), "bytes: d", d, "m", m, "N", N, "z_dim", t.dim)

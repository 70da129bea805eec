#!/usr/bin/env python3
"""Deviation of the diagonal Pade-p constraint from the reference's exp constraint, per BASELINE config.

On an exp-FEASIBLE synthetic trajectory (X_{k+1} = expm(dt_k G(u_k)) X_k exactly, the bench's input distribution without the
feasibility noise) the reference's residual x_{k+1} - expv(dt, G, x_k) [REF docs/src/concepts/index.md:21] is zero to
rounding, so max |B^-_p X_{k+1} - B^+_p X_k| IS the modelling difference of order p.  Runs the oracle (test infrastructure),
writes profiles/pade_vs_exp.json; bench.py and DESIGN.md quote the file (data, not code)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pade_oracle as po

out = {"what": "max |Pade-p residual| on an exp-feasible trajectory (bench input distribution, noise = 0), and max ||dt G(u_k)||_2",
       "generated_by": "scripts/pade_vs_exp.py (oracle/pade_oracle.py)", "configs": {}}
for cfg, N in ((1, 50), (2, 100), (3, 100)):
    so = po.config_system(cfg)
    G0, Gj = so.G_drift, np.array(so.G_drives)
    Z, lay = po.synthetic_trajectory(so, N, seed=20260929 + cfg, noise=0.0)
    nrm = max(np.linalg.norm(lay.dt(Z, k) * (G0 + np.tensordot(lay.u(Z, k), Gj, axes=1)), 2) for k in range(lay.K))
    row = {"N": N, "max_norm_dtG": float(nrm)}
    for p in (2, 4, 6, 8, 10):
        row["order_%d" % p] = float(np.abs(po.pade_residual(Z, lay, G0, Gj, p)).max())
    out["configs"]["config%d" % cfg] = row
    print(cfg, row)
with open(os.path.join(ROOT, "profiles", "pade_vs_exp.json"), "w") as f:
    json.dump(out, f, indent=1)

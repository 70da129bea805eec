"""Synthetic workloads of BASELINE.json (definitions: SURVEY.md section 8(d)).

Near-feasible iterates, like an interior-point iterate: X_1 = iso(I),
X_{k+1} = expm(dt G(u_k)) X_k + 1e-3 N(0,1);  u_k ~ 0.02 N(0,1) clipped to the drive bound;
dt = 0.1 (T ~ 10 ns);  PRNG numpy default_rng(seed)."""
import numpy as np
import scipy.linalg

from .quantum import MultiTransmonSystem, QuantumSystem, PAULIS
from .trajectory import NamedTrajectory


def config_system(config):
    if config == 1:
        return QuantumSystem(0.5 * PAULIS["Z"], [PAULIS["X"], PAULIS["Y"]], [1.0, 1.0])
    if config == 2:
        return MultiTransmonSystem([4.0, 4.1], [0.2, 0.2], [[0, 0.1], [0.1, 0]], levels_per_transmon=2, drive_bounds=0.1)
    if config in (3, 4, 5):
        return MultiTransmonSystem([4.0, 4.1, 4.2], [0.2, 0.21, 0.22], [[0, 0.01, 0.02], [0.01, 0, 0.03], [0.02, 0.03, 0]],
                                   levels_per_transmon=3, drive_bounds=0.1)  # fmt: skip
    raise ValueError("unknown config %r" % (config,))


def synthetic_trajectory(system, N, seed, dt=0.1, u_scale=0.02, noise=1e-3):
    """NamedTrajectory with components [Utilde, dt, t, u, du, ddu] (SmoothPulseProblem layout)."""
    rng = np.random.default_rng(seed)
    d, m = system.levels, system.n_drives
    clip = system.drive_bounds[0][1] if m else 1.0
    u = np.clip(u_scale * rng.standard_normal((N, m)), -clip, clip)
    du = 0.01 * rng.standard_normal((N, m))
    ddu = 0.01 * rng.standard_normal((N, m))
    Gj = system.G_drives_array()
    X = np.vstack([np.eye(d), np.zeros((d, d))])
    xs = np.empty((N, 2 * d * d))
    for k in range(N):
        xs[k] = X.T.reshape(-1)
        if k + 1 < N:
            Gk = system.G_drift + np.tensordot(u[k], Gj, axes=1)
            X = scipy.linalg.expm(dt * Gk) @ X + noise * rng.standard_normal(X.shape)
    comps = {"Ũ⃗": xs.T, "Δt": np.full((1, N), dt), "t": (dt * np.arange(N))[None], "u": u.T, "du": du.T, "ddu": ddu.T}
    return NamedTrajectory(comps, controls=("ddu", "Δt"), timestep="Δt")


def config4_members(first, count, indices=None):
    """Members first..first+count-1 (or the members `indices`: a rank's round-robin share) of BASELINE config 4: H_drift_i = H_drift + eps_i * 2 pi * sum_q a_q' a_q,
    eps_i ~ U(-1e-3, 1e-3) GHz from default_rng(2000 + i) -- a frequency-drift perturbation in the spirit of
    [REF docs/literate/robust_control.jl:82-83].  The drive Hamiltonians are shared."""
    from .quantum import annihilate, lift_operator

    base = config_system(3)
    lv = base.subsystem_levels
    a = annihilate(lv[0])
    num = sum(lift_operator(a.conj().T @ a, q, lv) for q in range(1, len(lv) + 1))
    out = []
    for i in (indices if indices is not None else range(first, first + count)):
        eps = np.random.default_rng(2000 + i).uniform(-1e-3, 1e-3)
        out.append(QuantumSystem(base.H_drift + eps * 2 * np.pi * num, base.H_drives, base.drive_bounds))
    return out


def synthetic_ensemble(systems, N, seed, dt=0.1, u_scale=0.02, noise=1e-3):
    """SamplingTrajectory-layout NamedTrajectory [Utilde1 .. UtildeM, dt, t, u, du, ddu] (shared controls; every member
    near ITS OWN exact rollout) [REF src/quantum/trajectories/sampling_trajectory.jl:207-237, sampling_problem.jl:346-376]."""
    rng = np.random.default_rng(seed)
    base = systems[0]
    d, m = base.levels, base.n_drives
    clip = base.drive_bounds[0][1] if m else 1.0
    u = np.clip(u_scale * rng.standard_normal((N, m)), -clip, clip)
    du = 0.01 * rng.standard_normal((N, m))
    ddu = 0.01 * rng.standard_normal((N, m))
    Gj = base.G_drives_array()
    comps = {}
    for i, s in enumerate(systems, 1):
        X = np.vstack([np.eye(d), np.zeros((d, d))])
        xs = np.empty((N, 2 * d * d))
        for k in range(N):
            xs[k] = X.T.reshape(-1)
            if k + 1 < N:
                X = scipy.linalg.expm(dt * (s.G_drift + np.tensordot(u[k], Gj, axes=1))) @ X + noise * rng.standard_normal(X.shape)
        comps["Ũ⃗%d" % i] = xs.T
    comps.update({"Δt": np.full((1, N), dt), "t": (dt * np.arange(N))[None], "u": u.T, "du": du.T, "ddu": ddu.T})
    return NamedTrajectory(comps, controls=("ddu", "Δt"), timestep="Δt")

"""Shared test helpers (oracle-side systems for the committed reference trajectories)."""
import numpy as np

from oracle import pade_oracle as po


def ref_case(name, meta):
    """(list of oracle systems, Layout, per-member x_offs) for a reference-solved trajectory."""
    m = meta["reference_trajectories"][name]
    d, nd, N, zd, M = m["d"], m["m"], m["N"], m["z_dim"], m["n_members"]
    Hd, Hs = 0.5 * po.PAULIS["Z"], [po.PAULIS["X"], po.PAULIS["Y"]]
    if name == "two_qubit_zoh":
        systems = [po.config_system(2)]
    elif name == "multilevel_transmon":
        systems = [po.transmon_system(levels=5, delta=0.2, drive_bounds=[0.2, 0.2])]
    elif name == "first_gate":
        systems = [po.config_system(1)]
    elif name == "sampling_robust":
        systems = [po.quantum_system(s * Hd, Hs, [1.0, 1.0]) for s in (1.0, 1.05, 0.95)]
    elif name == "robust_sampling":
        systems = [po.quantum_system(s * Hd, Hs, [1.0, 1.0]) for s in (0.9, 0.95, 1.0, 1.05, 1.1)]
    else:
        raise KeyError(name)
    xd = 2 * d * d
    if M == 1:
        lay = po.Layout.smooth_pulse(d, nd, N)
        assert lay.z_dim == zd
    else:  # [U1..UM, dt, t, u]
        lay = po.Layout(d=d, m=nd, N=N, z_dim=zd, x_off=0, u_off=M * xd + 2, dt_off=M * xd)
    return systems, lay, [i * xd for i in range(M)]


def traj_from_Z(pa, Z, lay, n_members=1):
    """Wrap an [N, z_dim] knot array as a product-side NamedTrajectory with the template's component names."""
    comps = {}
    xd = lay.x_dim
    if n_members == 1:
        comps["Ũ⃗"] = Z[:, lay.x_off : lay.x_off + xd].T
    else:
        for i in range(n_members):
            comps["Ũ⃗%d" % (i + 1)] = Z[:, i * xd : (i + 1) * xd].T
    comps["Δt"] = Z[:, lay.dt_off][None]
    comps["t"] = Z[:, lay.dt_off + 1][None]
    m = lay.m
    comps["u"] = Z[:, lay.u_off : lay.u_off + m].T
    rest = lay.z_dim - (lay.u_off + m)
    if rest >= 2 * m and m:
        comps["du"] = Z[:, lay.u_off + m : lay.u_off + 2 * m].T
        comps["ddu"] = Z[:, lay.u_off + 2 * m : lay.u_off + 3 * m].T
    t = pa.NamedTrajectory(comps, controls=("u", "Δt"), timestep="Δt")
    assert t.dim == lay.z_dim and np.array_equal(t.datavec, Z.reshape(-1))
    return t

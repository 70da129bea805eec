"""Knot-major trajectory container: the layout contract of the evaluator.

Mirrors the part of NamedTrajectories.jl's ``NamedTrajectory`` that Piccolo's
integrators read: a flat ``datavec`` holding a
``dim x N`` column-major matrix (knot k is the contiguous slice
``datavec[k*dim:(k+1)*dim]``), named component ranges inside a knot, and
``global_data`` appended after the knots in the NLP variable vector
[REF src/control/integrators.jl:781-783].  Component order of the problem
templates is ``[Utilde(s), dt, t, u, du, ddu]``
[REF src/quantum/trajectories/named_trajectory_conversion.jl:321,339-351;
 src/quantum/trajectories/sampling_trajectory.jl:207-237;
 src/control/templates/smooth_pulse_problem.jl:187-201].

Ranges are 0-based Python ``range`` objects (Julia's are 1-based ``UnitRange``).
"""
from collections import OrderedDict

import numpy as np

from .quantum import operator_to_iso_vec

STATE = "Ũ⃗"  # :Ũ⃗  (state_name of a UnitaryTrajectory)
KET = "ψ̃"  # :ψ̃  (state_name of a KetTrajectory)
DENSITY = "ρ⃗̃"  # :ρ⃗̃  (state_name of a DensityTrajectory: compact isomorphism, levels^2 reals)
TIMESTEP = "Δt"


class NamedTrajectory:
    def __init__(self, components, controls=(), timestep=TIMESTEP, bounds=None, initial=None, final=None, goal=None,
                 global_data=None):  # fmt: skip
        comps = OrderedDict()
        N = None
        for name, arr in components.items():
            a = np.atleast_2d(np.asarray(arr, dtype=np.float64))
            if N is None:
                N = a.shape[1]
            if a.shape[1] != N:
                raise ValueError("component %s has %d knots, expected %d" % (name, a.shape[1], N))
            comps[name] = a
        if N is None or N < 1:
            raise ValueError("a trajectory needs at least one component")
        self.N = N
        self.names = tuple(comps)
        self.dims = OrderedDict((k, v.shape[0]) for k, v in comps.items())
        self.components = OrderedDict()
        off = 0
        for k, v in comps.items():
            self.components[k] = range(off, off + v.shape[0])
            off += v.shape[0]
        self.dim = off
        # knot-major flat buffer: data[:, k] contiguous
        self.datavec = np.concatenate([v for v in comps.values()], axis=0).T.reshape(-1).copy()
        self.control_names = tuple(controls)
        self.timestep = timestep
        self.state_names = tuple(n for n in self.names if n not in self.control_names)
        self.bounds = dict(bounds or {})
        self.initial = dict(initial or {})
        self.final = dict(final or {})
        self.goal = dict(goal or {})
        gd = OrderedDict((k, np.atleast_1d(np.asarray(v, dtype=np.float64))) for k, v in (global_data or {}).items())
        self.global_names = tuple(gd)
        self.global_components = OrderedDict()
        goff = 0
        for k, v in gd.items():
            self.global_components[k] = range(goff, goff + v.size)
            goff += v.size
        self.global_dim = goff
        self.global_data = np.concatenate(list(gd.values())) if gd else np.zeros(0)

    # -- views --------------------------------------------------------------------------------
    @property
    def data(self):
        """dim x N view of ``datavec`` (column k = knot k)."""
        return self.datavec.reshape(self.N, self.dim).T

    def __getitem__(self, name):
        r = self.components[name]
        return self.data[r.start : r.stop, :]

    def knot(self, k):
        return self.datavec[k * self.dim : (k + 1) * self.dim]

    def variables(self):
        """[datavec; global_data] -- the NLP decision vector."""
        return np.concatenate((self.datavec, self.global_data))

    def update(self, datavec):
        datavec = np.asarray(datavec, dtype=np.float64).reshape(-1)
        if datavec.size != self.dim * self.N:
            raise ValueError("datavec has %d entries, expected %d" % (datavec.size, self.dim * self.N))
        self.datavec[:] = datavec

    def copy(self):
        import copy as _copy

        t = _copy.copy(self)
        t.datavec = self.datavec.copy()
        t.global_data = self.global_data.copy()
        return t

    def __repr__(self):
        return "NamedTrajectory(N=%d, dim=%d, components=%s)" % (self.N, self.dim, dict(self.dims))


def add_control_derivatives(traj, n_derivs, control_name="u"):
    """Append ``du``, ``ddu`` ... components (finite-difference initialised), as
    ``add_control_derivatives(traj, 2)`` does in the templates
    [REF src/control/templates/smooth_pulse_problem.jl:196-201]."""
    comps = OrderedDict((k, traj[k].copy()) for k in traj.names)
    dt = traj[traj.timestep][0]
    prev, name = comps[control_name], control_name
    for _ in range(n_derivs):
        name = "d" + name
        der = np.zeros_like(prev)
        der[:, :-1] = (prev[:, 1:] - prev[:, :-1]) / dt[:-1]
        der[:, -1] = der[:, -2] if traj.N > 1 else 0.0
        comps[name] = der
        prev = der
    return NamedTrajectory(comps, controls=(name, traj.timestep), timestep=traj.timestep, bounds=traj.bounds,
                           initial=traj.initial, final=traj.final, goal=traj.goal)  # fmt: skip


def unitary_trajectory(system, controls, times, U_goal, states=None, n_derivs=2, state_name=STATE):
    """``NamedTrajectory(qtraj::UnitaryTrajectory, N)`` + ``add_control_derivatives``:
    components ``[Utilde, dt, t, u, du, ddu]``.  ``states`` (list of d x d unitaries per
    knot) defaults to the identity at every knot (the reference samples an ODE rollout)."""
    times = np.asarray(times, float)
    N = times.size
    u = np.asarray(controls, float).reshape(system.n_drives, N)
    d = system.levels
    if states is None:
        states = [np.eye(d)] * N
    X = np.stack([operator_to_iso_vec(U) for U in states], axis=1)
    dts = np.diff(times)
    dts = np.concatenate((dts, dts[-1:])) if N > 1 else np.ones(1)
    comps = OrderedDict([(state_name, X), (TIMESTEP, dts[None, :]), ("t", times[None, :]), ("u", u)])
    traj = NamedTrajectory(
        comps,
        controls=(TIMESTEP, "u"),
        timestep=TIMESTEP,
        bounds={state_name: (-np.ones(2 * d * d), np.ones(2 * d * d)), "u": system.drive_bounds},
        initial={state_name: operator_to_iso_vec(np.eye(d))},
        goal={state_name: operator_to_iso_vec(U_goal)},
    )
    return add_control_derivatives(traj, n_derivs) if n_derivs else traj


def sampling_trajectory(systems, controls, times, U_goal, states=None, state_name=STATE):
    """``NamedTrajectory(::SamplingTrajectory, N)``: components
    ``[Utilde1 .. UtildeM, dt, t, u]`` -- one state copy per ensemble member, shared
    controls [REF src/quantum/trajectories/sampling_trajectory.jl:181-238]."""
    times = np.asarray(times, float)
    N = times.size
    base = systems[0]
    u = np.asarray(controls, float).reshape(base.n_drives, N)
    d = base.levels
    if states is None:
        states = [np.eye(d)] * N
    X = np.stack([operator_to_iso_vec(U) for U in states], axis=1)
    dts = np.diff(times)
    dts = np.concatenate((dts, dts[-1:]))
    comps = OrderedDict()
    for i in range(1, len(systems) + 1):
        comps["%s%d" % (state_name, i)] = X.copy()
    comps[TIMESTEP] = dts[None, :]
    comps["t"] = times[None, :]
    comps["u"] = u
    names = ["%s%d" % (state_name, i) for i in range(1, len(systems) + 1)]
    return NamedTrajectory(
        comps,
        controls=(TIMESTEP, "u"),
        timestep=TIMESTEP,
        initial={nm: operator_to_iso_vec(np.eye(d)) for nm in names},
        goal={nm: operator_to_iso_vec(U_goal) for nm in names},
    )


def ket_trajectory(system, controls, times, psi_init, psi_goal, states=None, n_derivs=2, state_name=KET):
    """``NamedTrajectory(qtraj::KetTrajectory, N)``: components ``[psitilde, dt, t, u, du, ddu]`` with
    ``psitilde = [Re psi; Im psi]`` [REF src/quantum/primitives/isomorphisms.jl:55;
    src/quantum/trajectories/named_trajectory_conversion.jl]."""
    from .quantum import ket_to_iso

    times = np.asarray(times, float)
    N = times.size
    u = np.asarray(controls, float).reshape(system.n_drives, N)
    if states is None:
        states = [np.asarray(psi_init, complex)] * N
    X = np.stack([ket_to_iso(s_) for s_ in states], axis=1)
    dts = np.diff(times)
    dts = np.concatenate((dts, dts[-1:])) if N > 1 else np.ones(1)
    comps = OrderedDict([(state_name, X), (TIMESTEP, dts[None, :]), ("t", times[None, :]), ("u", u)])
    traj = NamedTrajectory(comps, controls=(TIMESTEP, "u"), timestep=TIMESTEP,
                           initial={state_name: ket_to_iso(psi_init)}, goal={state_name: ket_to_iso(psi_goal)})  # fmt: skip
    return add_control_derivatives(traj, n_derivs) if n_derivs else traj


def density_trajectory(system, controls, times, rho_init, rho_goal, states=None, n_derivs=2, state_name=DENSITY):
    """``NamedTrajectory(qtraj::DensityTrajectory, N)``: components ``[rhotilde, dt, t, u, du, ddu]`` with
    ``rhotilde = density_to_compact_iso(rho)`` (levels^2 reals) [REF named_trajectory_conversion.jl:540-600]."""
    from .quantum import density_to_compact_iso

    times = np.asarray(times, float)
    N = times.size
    u = np.asarray(controls, float).reshape(system.n_drives, N)
    if states is None:
        states = [np.asarray(rho_init, complex)] * N
    X = np.stack([density_to_compact_iso(r) for r in states], axis=1)
    dts = np.diff(times)
    dts = np.concatenate((dts, dts[-1:])) if N > 1 else np.ones(1)
    comps = OrderedDict([(state_name, X), (TIMESTEP, dts[None, :]), ("t", times[None, :]), ("u", u)])
    traj = NamedTrajectory(comps, controls=(TIMESTEP, "u"), timestep=TIMESTEP,
                           initial={state_name: density_to_compact_iso(rho_init)}, goal={state_name: density_to_compact_iso(rho_goal)})  # fmt: skip
    return add_control_derivatives(traj, n_derivs) if n_derivs else traj

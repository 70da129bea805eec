"""Host-side mirror of the objective terms of the reference's unitary problem templates, evaluated ON THE GPU through
the C ABI (``pcl_set_goal[_subspace]``, ``pcl_set_weights``, ``pcl_add_regularizer``, ``pcl_objective[_dev]``):

  UnitaryInfidelityObjective(U_goal, name, traj; Q)        src/control/objectives.jl:347-356   (Q * |1 - F(U_N)|)
  ... with an EmbeddedOperator goal (subspace fidelity)     src/control/objectives.jl:339-345
  QuadraticRegularizer(name, traj, R) [EXT DirectTrajOpt]   src/control/templates/smooth_pulse_problem.jl:249-251
  sum_i (w_i Q) l_i + regularisers (SamplingProblem)        src/control/templates/sampling_problem.jl:381-387

Terms are small records combined with ``+`` (as the reference combines ``AbstractObjective``s); ``bind`` attaches the sum
to an integrator's context, after which ``value_and_gradient(traj)`` is one call into the library.  Nothing here computes
an objective on the host.
"""
import numpy as np

from .quantum import compact_iso_to_density, operator_to_iso_vec

__all__ = ["EmbeddedOperator", "UnitaryInfidelityObjective", "QuadraticRegularizer", "Objective", "get_subspace_indices",
           "KetInfidelityObjective", "CoherentKetInfidelityObjective", "DensityMatrixInfidelityObjective", "DensityMatrixPureStateInfidelityObjective"]  # fmt: skip


def get_subspace_indices(subspaces, subsystem_levels):
    """0-based indices of the product-basis states whose every subsystem level lies in that subsystem's subspace
    (``subspaces``: one 0-based level list per subsystem) [REF src/quantum/operators/embedded_operators.jl:352-364]."""
    import itertools

    return [flat for flat, lv in enumerate(itertools.product(*[range(L) for L in subsystem_levels]))
            if all(l in sub for l, sub in zip(lv, subspaces))]  # fmt: skip


class EmbeddedOperator:
    """``EmbeddedOperator(subspace_operator, subspace, subsystem_levels)`` [REF embedded_operators.jl:70-89]; ``subspace``
    holds 0-based indices here."""

    def __init__(self, subspace_operator, subspace, subsystem_levels):
        levels = [subsystem_levels] if np.isscalar(subsystem_levels) else list(subsystem_levels)
        self.subspace = [int(i) for i in subspace]
        self.subsystem_levels = levels
        n = int(np.prod(levels))
        op = np.asarray(subspace_operator, dtype=complex)
        if op.shape != (len(self.subspace),) * 2:
            raise ValueError("subspace operator is %r, subspace has %d indices" % (op.shape, len(self.subspace)))
        self.operator = np.zeros((n, n), dtype=complex)
        self.operator[np.ix_(self.subspace, self.subspace)] = op

    def unembed(self):
        return self.operator[np.ix_(self.subspace, self.subspace)]


class _Term:
    def __add__(self, other):
        return Objective(_terms(self) + _terms(other))

    __radd__ = lambda self, other: self if other == 0 else NotImplemented


def _terms(x):
    return list(x.terms) if isinstance(x, Objective) else [x]


class UnitaryInfidelityObjective(_Term):
    """``Q * |1 - F|`` on the terminal state(s) ``names`` (one name, or the member states of an ensemble with optional
    ``weights`` -- the SamplingProblem sum)."""

    def __init__(self, U_goal, names, traj=None, Q=100.0, weights=None):
        self.goal, self.Q = U_goal, float(Q)
        self.names = [names] if isinstance(names, str) else list(names)
        self.weights = None if weights is None else np.asarray(weights, dtype=np.float64)


def _ket_rows(goal):
    """<g|psi> = a'x + i b'x for x = [Re psi; Im psi]."""
    g = np.asarray(goal, dtype=complex).reshape(-1)
    return np.concatenate([g.real, g.imag]), np.concatenate([-g.imag, g.real])


class _FormTerm(_Term):
    """A terminal loss ``Q |1 - F(x)|`` with ``F(x) = c'x + sum_r (A_r'x)^2`` (``pcl_set_goal_form``): ``form(x_dim, n_members)`` returns
    ``(scope, A, c)`` -- scope 0: one term per member named in ``names``, 1: ONE term over all of them concatenated."""

    Q = 100.0
    names = ()
    weights = None


class KetInfidelityObjective(_FormTerm):
    """``Q |1 - |<goal|psi_N>|^2|`` on the ket component(s) ``names`` [REF src/control/objectives.jl:24-60] (several names: the members
    of an ensemble, one term each with optional ``weights`` -- the SamplingProblem sum)."""

    def __init__(self, psi_goal, names, traj=None, Q=100.0, weights=None):
        self.goal, self.Q = np.asarray(psi_goal, dtype=complex), float(Q)
        self.names = [names] if isinstance(names, str) else list(names)
        self.weights = None if weights is None else np.asarray(weights, dtype=np.float64)

    def form(self, x_dim, n_members):
        a, b = _ket_rows(self.goal)
        if a.size != x_dim:
            raise ValueError("goal ket has %d amplitudes, the state component %d reals" % (self.goal.size, x_dim))
        return 0, np.stack([a, b]), None


class CoherentKetInfidelityObjective(_FormTerm):
    """``Q |1 - |sum_i w_i <g_i|psi_i> / sum_i w_i|^2|`` over the kets ``names`` -- ONE term: the overlaps must share a phase
    [REF src/control/objectives.jl:96-200].  Uniform weights (or none) are the unweighted mean."""

    def __init__(self, psi_goals, names, traj=None, Q=100.0, weights=None):
        self.goals = [np.asarray(g, dtype=complex) for g in psi_goals]
        self.names, self.Q = list(names), float(Q)
        if len(self.goals) != len(self.names):
            raise ValueError("number of names must match number of goals")
        w = None if weights is None else np.asarray(weights, dtype=np.float64)
        if w is not None:
            if w.size != len(self.goals) or (w < 0).any() or w.sum() <= 0:
                raise ValueError("weights: one non-negative weight per state, not all zero")
            w = None if np.all(w == w[0]) else w / w.sum()  # [REF objectives.jl:137-143]: uniform weights are the unweighted path
        self.coherent_weights = w
        self.weights = None  # (no per-term weights: there is one term)

    def form(self, x_dim, n_members):
        n = len(self.goals)
        if n != n_members:
            raise ValueError("the coherent term names %d kets, the integrator list evaluates %d" % (n, n_members))
        w = np.full(n, 1.0 / n) if self.coherent_weights is None else self.coherent_weights
        rows = [_ket_rows(g) for g in self.goals]
        return 1, np.stack([np.concatenate([w[i] * rows[i][0] for i in range(n)]), np.concatenate([w[i] * rows[i][1] for i in range(n)])]), None


class DensityMatrixInfidelityObjective(_FormTerm):
    """``Q |1 - Re tr(rho_N rho_goal)|`` on the compact-iso density component ``name`` [REF src/control/objectives.jl:387-411]: linear in
    the state, ``c_e = Re tr(E_e rho_goal)`` with ``E_e`` the density the e-th unit vector stands for."""

    def __init__(self, names, rho_goal, traj=None, Q=100.0, weights=None):
        self.rho_goal, self.Q = np.asarray(rho_goal, dtype=complex), float(Q)
        self.names = [names] if isinstance(names, str) else list(names)
        self.weights = None if weights is None else np.asarray(weights, dtype=np.float64)

    def form(self, x_dim, n_members):
        if self.rho_goal.shape != (int(round(np.sqrt(x_dim))),) * 2:
            raise ValueError("goal density is %r, the compact state has %d entries" % (self.rho_goal.shape, x_dim))
        c = np.array([np.trace(compact_iso_to_density(e) @ self.rho_goal).real for e in np.eye(x_dim)])
        return 0, None, c


class DensityMatrixPureStateInfidelityObjective(DensityMatrixInfidelityObjective):
    """``Q |1 - Re <psi|rho_N|psi>|`` [REF src/control/objectives.jl:413-435]."""

    def __init__(self, names, psi_goal, traj=None, Q=100.0, weights=None):
        psi = np.asarray(psi_goal, dtype=complex).reshape(-1)
        super().__init__(names, np.outer(psi, psi.conj()), traj, Q, weights)


class QuadraticRegularizer(_Term):
    """``1/2 sum_k dt_k^p sum_i R_i v_{k,i}^2`` on component ``name``; ``R`` a scalar or one weight per entry.
    ``dt_power`` = 2 is the DirectTrajOpt / QuantumCollocation form (r = dt v), 0 the plain knot-point form."""

    def __init__(self, name, traj, R, dt_power=2):
        self.name, self.dt_power = name, int(dt_power)
        self.off, self.dim = traj.components[name].start, len(traj.components[name])
        self.R = np.broadcast_to(np.asarray(R, dtype=np.float64), (self.dim,)).copy()


class Objective:
    def __init__(self, terms):
        self.terms = list(terms)
        self._ctx = None
        self._bound = []
        self._Q = 0.0

    def __add__(self, other):
        return Objective(self.terms + _terms(other))

    def bind(self, B):
        """Attach to the context(s) of integrator ``B``: a single integrator, or the integrator list of an ensemble.

        A list whose members share ONE batched context (per-member drifts only) binds that context: the weighted
        SamplingProblem sum is formed on the device.  A list of independent contexts (members that differ in their drive
        generators too) binds every member's context with its own weight and sums on the host; the regularisers -- terms
        of the shared controls -- are registered once, on the first member."""
        members = list(B) if isinstance(B, (list, tuple)) else [B]
        inf = [t for t in self.terms if isinstance(t, (UnitaryInfidelityObjective, _FormTerm))]
        if len(inf) > 1:
            raise NotImplementedError("one terminal infidelity term per problem")
        cores = {id(b.ensemble) for b in members if hasattr(b, "ensemble")}
        shared = len(cores) == 1 and all(hasattr(b, "ensemble") for b in members)
        if len(members) > 1 and not shared and any(hasattr(b, "ensemble") for b in members):
            raise ValueError("the integrator list mixes members of different ensembles")
        if shared:
            # any member (or sub-list) of an ensemble binds the ONE batched context all members share; an infidelity term then has to
            # name every member's state (checked below) -- regulariser-only objectives need no more than the context
            core = members[0].ensemble
            if inf and len(members) != core.M:
                raise ValueError("the infidelity term needs the whole ensemble: %d integrators of %d members" % (len(members), core.M))
            ctxs = [core.ctx]
        else:
            ctxs = [b.ctx for b in members]
        if inf and all(hasattr(b, "x_names") for b in members):  # (a multistart context carries one state name for all its seeds)
            t = inf[0]
            have = [nm for b in members for nm in (b.x_names if not hasattr(b, "ensemble") else [b.x_name])]
            if list(t.names) != have:
                raise ValueError("the infidelity term names the states %r, the integrators evaluate %r" % (list(t.names), have))
            if t.weights is not None and t.weights.size != len(have):
                raise ValueError("expected %d weights, got %d" % (len(have), t.weights.size))
        self._bound = []
        self._Q = 0.0
        for i, ctx in enumerate(ctxs):
            ctx.clear_regularizers()
            if i == 0:
                for t in self.terms:
                    if isinstance(t, QuadraticRegularizer):
                        ctx.add_regularizer(t.off, t.dim, t.R, t.dt_power)
            w_host = 1.0
            if inf and isinstance(inf[0], _FormTerm):
                t = inf[0]
                if len(ctxs) != 1:
                    raise NotImplementedError("ket / density losses on members with contexts of their own")
                scope, A, c = t.form(ctx.x_dim, ctx.batch)
                ctx.set_goal_form(scope, A, c)
                ctx.set_weights(t.weights)
                self._Q = t.Q
            elif inf:
                t = inf[0]
                if isinstance(t.goal, EmbeddedOperator):
                    ctx.set_goal_subspace(operator_to_iso_vec(t.goal.unembed()), t.goal.subspace)
                else:
                    ctx.set_goal(operator_to_iso_vec(np.asarray(t.goal, dtype=complex)))
                if len(ctxs) == 1:
                    ctx.set_weights(t.weights)
                else:  # one member per context: its weight multiplies the member's infidelity on the host
                    ctx.set_weights(None)
                    w_host = 1.0 if t.weights is None else float(t.weights[i])
                self._Q = t.Q
            if inf or i == 0:  # (a context that carries no term -- regularisers live on the first one -- is not evaluated)
                self._bound.append((ctx, w_host))
        self._ctx = ctxs[0]
        return self

    def value_and_gradient(self, traj_or_Z, want_grad=True):
        """(J, dJ/dz) for the trajectory's variable vector (host buffers; one value per seed for a multistart context)."""
        if self._ctx is None:
            raise RuntimeError("bind the objective to an integrator first")
        Z = traj_or_Z.datavec if hasattr(traj_or_Z, "datavec") else traj_or_Z
        if len(self._bound) == 1:
            v, g = self._ctx.objective(Z, self._Q, want_grad)
            return (float(v[0]) if v.size == 1 else v), g
        # independent contexts: member i contributes w_i Q |1 - F_i| (+ the regularisers, registered on member 0 only)
        total, grad = 0.0, None
        for ctx, w in self._bound:
            v, g = ctx.objective(Z, self._Q * w, want_grad)
            total += float(v[0])
            if want_grad:
                grad = g.copy() if grad is None else grad + g
        return total, grad

    def hessian_structure(self):
        """(rows, cols) of the objective's Hessian values (each entry once, row >= col; the context's index base)."""
        if len(self._bound) != 1:
            raise NotImplementedError("Hessian of an objective spread over several contexts")
        return self._ctx.objective_hess_structure()

    def hessian(self, traj_or_Z, sigma=1.0):
        """sigma * grad^2 J in the order of ``hessian_structure`` -- what ``eval_hessian_lagrangian`` adds to ``pcl_hess``'s term."""
        if len(self._bound) != 1:
            raise NotImplementedError("Hessian of an objective spread over several contexts")
        Z = traj_or_Z.datavec if hasattr(traj_or_Z, "datavec") else traj_or_Z
        return self._ctx.objective_hess(Z, self._Q, sigma)

    def step_dev(self, Z_dev, value_dev, grad_dev, delta_dev, vals_dev, payload_dev, lam_dev=None):
        """A rank's whole step of a sharded ensemble on the device: objective value and gradient (as ``value_and_gradient_dev``), the
        members' residuals and Jacobian values and the reduce payload (as the context's ``eval_jac_merit_dev``) -- two launches."""
        if len(self._bound) != 1:
            raise NotImplementedError("device-resident step of an ensemble whose members have their own contexts")
        self._ctx.eval_jac_merit_objective_dev(Z_dev, lam_dev, delta_dev, vals_dev, payload_dev, self._Q, value_dev, grad_dev)

    def value_and_gradient_dev(self, Z_dev, value_dev, grad_dev=None):
        if len(self._bound) != 1:
            raise NotImplementedError("device-resident objective of an ensemble whose members have their own contexts: use value_and_gradient")
        self._ctx.objective_dev(Z_dev, self._Q, value_dev, grad_dev)

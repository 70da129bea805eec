"""Host-side mirror of the reference's integrator plug-in surface for the unitary
Pade collocation constraint, backed by libpiccolo_hip.so (HIP kernels, gfx950).

What the reference calls (Piccolo.jl v2.0.2; [EXT] = DirectTrajOpt.jl):

  BilinearIntegrator(qtraj::UnitaryTrajectory, N)          src/control/integrators.jl:35-51
  BilinearIntegrator(qtraj::SamplingTrajectory, N)         src/control/integrators.jl:134-162
  B.dim, B.x_dim, B.x_name / B.x_names, B.f(x', x, u, dt)  src/control/integrators.jl:307-309,525,552
  evaluate!(delta, B, traj) [EXT]                          src/control/integrators.jl:311,777
  eval_jacobian(B, traj) [EXT]  -> (B.dim, traj.dim*traj.N + traj.global_dim)   :780-783
  hessian_structure / Hessian of the Lagrangian [EXT]      test/aqua.jl:6-9

Here the same names exist as Python callables (``evaluate_`` for ``evaluate!``).
All arithmetic happens on the GPU through the C ABI; nothing in this module
computes a residual or a Jacobian on the host, and nothing falls back to a CPU
implementation when the library or the device is missing.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import PCL_BATCH_MEMBERS, PCL_BATCH_TRAJ, PclError
from .trajectory import STATE, TIMESTEP, NamedTrajectory

__all__ = [
    "HipPadeIntegrator", "HipPadeMemberIntegrator", "HipPadeMultistart", "DerivativeIntegrator", "BilinearIntegrator", "evaluate_", "eval_jacobian",
    "jacobian_structure", "hessian_structure", "eval_hessian_of_lagrangian", "PclError",
]  # fmt: skip


def _colmajor(A):
    """numpy [i, j] matrix (or stack of matrices) -> flat column-major float64 buffer."""
    A = np.asarray(A, dtype=np.float64)
    return np.ascontiguousarray(np.swapaxes(A, -1, -2)).reshape(-1)


def _ptr(a):
    """Address of a host numpy array or a torch (device) tensor."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if a.dtype != np.float64 or not a.flags.c_contiguous:
            raise TypeError("expected a C-contiguous float64 array")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):  # torch tensor
        if str(a.dtype) != "torch.float64" or not a.is_contiguous():
            raise TypeError("expected a contiguous float64 tensor")
        return a.data_ptr()
    raise TypeError("unsupported buffer type %r" % type(a))


class _PclContext:
    """Owns one ``pcl_ctx`` (one GPU, one stream)."""

    def __init__(self, *, d, m, N, z_dim, u_off, dt_off, x_offs, G0, Gj, batch, batch_mode, per_member_G0=False,
                 global_dim=0, device=0, index_base=0, pade_order=4, state_cols=0):  # fmt: skip
        self._L = _lib.load()
        self._h = None
        n = d if state_cols == _lib.PCL_STATE_VECTOR else 2 * d  # PCL_STATE_VECTOR: general d x d generator, one column
        g0 = _colmajor(G0)
        gj = _colmajor(Gj) if m else np.zeros(1)
        if g0.size != n * n * (batch if per_member_G0 else 1):
            raise ValueError("G0 has %d entries, expected %d" % (g0.size, n * n * (batch if per_member_G0 else 1)))
        if m and gj.size != m * n * n:
            raise ValueError("Gj has %d entries, expected %d" % (gj.size, m * n * n))
        xo = np.ascontiguousarray(np.asarray(x_offs, dtype=np.int32).reshape(-1))
        desc = _lib.pcl_desc(
            struct_size=ctypes.sizeof(_lib.pcl_desc), d=d, n_drives=m, N=N, z_dim=z_dim, u_off=u_off, dt_off=dt_off,
            batch=batch, batch_mode=batch_mode, pade_order=pade_order, device_id=device, index_base=index_base,
            per_member_G0=int(bool(per_member_G0)), state_cols=state_cols, global_dim=global_dim,
            G0=g0.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            Gj=gj.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            x_offs=xo.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        )  # fmt: skip
        h = ctypes.c_void_p()
        rc = self._L.pcl_create(ctypes.byref(desc), ctypes.byref(h))
        if rc != 0:
            raise PclError(rc, (self._L.pcl_last_error(None) or b"").decode())
        self._h = h
        self.d, self.n, self.m, self.N, self.K, self.z_dim = d, n, m, N, N - 1, z_dim
        self.batch, self.batch_mode = batch, batch_mode
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._chk(self._L.pcl_constraint_dim(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        self.x_dim, self.n_rows, self.n_cols = a.value, b.value, c.value
        self._chk(self._L.pcl_jac_nnz(h, ctypes.byref(a), ctypes.byref(b)))
        self.jac_nnz, self.jac_per = a.value, b.value
        self._chk(self._L.pcl_hess_nnz(h, ctypes.byref(a), ctypes.byref(b)))
        self.hess_nnz, self.hess_per = a.value, b.value
        self._chk(self._L.pcl_jac_compact_nnz(h, ctypes.byref(a), ctypes.byref(b)))
        self.compact_nnz, self.compact_per = a.value, b.value
        self.z_len = z_dim * N * (batch if batch_mode == PCL_BATCH_TRAJ else 1)
        self.window = (0, batch)

    def _chk(self, rc):
        if rc != 0:
            raise PclError(rc, (self._L.pcl_last_error(self._h) or b"").decode())

    def set_order_policy(self, dt_max, u_max, tol=1e-10):
        """The smallest diagonal Pade order whose deviation from the reference's exp constraint, ``kappa_q theta^(2q+1)`` with
        ``theta = dt_max max_{|u_l| <= u_max_l} |G_drift + sum_l u_l G_l|_2`` (the maximum over the box of controls), is below ``tol``; becomes the context's order.  Returns it."""
        um = np.ascontiguousarray(np.broadcast_to(np.abs(np.asarray(u_max, dtype=np.float64)), (max(self.m, 1),)))
        out = ctypes.c_int32()
        self._L.pcl_set_order_policy.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_int32)]
        self._chk(self._L.pcl_set_order_policy(self._h, float(dt_max), um.ctypes.data, float(tol), ctypes.byref(out)))
        return out.value

    def set_order_from_trajectory(self, Z, tol=1e-10):
        """The same decision from a trajectory on the host (``theta = 1.5 max_k |dt_k G(u_k)|``): what a constructor calls when the
        trajectory carries no bounds on the drives and the timestep.  Returns the order."""
        Z = self._z(Z)
        out = ctypes.c_int32()
        self._L.pcl_set_order_from_trajectory.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_int32)]
        self._chk(self._L.pcl_set_order_from_trajectory(self._h, _ptr(Z), float(tol), ctypes.byref(out)))
        return out.value

    @property
    def order_tol_met(self):
        """False when the order policy had to settle for order 10 with its bound above the tolerance."""
        return bool(self.get_option("order_tol_met"))

    @property
    def pade_order(self):
        """The order in use (0: a context created with ``pade_order=0`` that has not seen a policy or a trajectory yet)."""
        return self.get_option("pade_order")

    def close(self):
        if getattr(self, "_h", None):
            self._L.pcl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- structure ----------------------------------------------------------------------------
    def jac_structure(self, dtype=np.int64):
        rows, cols = np.empty(self.jac_nnz, dtype), np.empty(self.jac_nnz, dtype)
        if dtype == np.int32:
            p = ctypes.POINTER(ctypes.c_int32)
            self._chk(self._L.pcl_jac_structure(self._h, rows.ctypes.data_as(p), cols.ctypes.data_as(p)))
        else:
            p = ctypes.POINTER(ctypes.c_int64)
            self._chk(self._L.pcl_jac_structure_i64(self._h, rows.ctypes.data_as(p), cols.ctypes.data_as(p)))
        return rows, cols

    def hess_structure(self, dtype=np.int64):
        rows, cols = np.empty(self.hess_nnz, dtype), np.empty(self.hess_nnz, dtype)
        if dtype == np.int32:
            p = ctypes.POINTER(ctypes.c_int32)
            self._chk(self._L.pcl_hess_structure(self._h, rows.ctypes.data_as(p), cols.ctypes.data_as(p)))
        else:
            p = ctypes.POINTER(ctypes.c_int64)
            self._chk(self._L.pcl_hess_structure_i64(self._h, rows.ctypes.data_as(p), cols.ctypes.data_as(p)))
        return rows, cols

    # -- host-pointer calls ---------------------------------------------------------------------
    def _z(self, Z):
        Z = np.ascontiguousarray(Z, dtype=np.float64).reshape(-1)
        if Z.size != self.z_len:
            raise ValueError("trajectory buffer has %d entries, expected %d" % (Z.size, self.z_len))
        return Z

    def eval(self, Z, delta=None):
        Z = self._z(Z)
        delta = np.empty(self.n_rows) if delta is None else delta
        self._chk(self._L.pcl_eval(self._h, _ptr(Z), _ptr(delta)))
        return delta

    def eval_jac(self, Z, delta=None, vals=None):
        Z = self._z(Z)
        delta = np.empty(self.n_rows) if delta is None else delta
        vals = np.empty(self.jac_nnz) if vals is None else vals
        self._chk(self._L.pcl_eval_jac(self._h, _ptr(Z), _ptr(delta), _ptr(vals)))
        return delta, vals

    def jac(self, Z, vals=None):
        Z = self._z(Z)
        vals = np.empty(self.jac_nnz) if vals is None else vals
        self._chk(self._L.pcl_jac(self._h, _ptr(Z), _ptr(vals)))
        return vals

    def hess(self, Z, mu, vals=None):
        Z = self._z(Z)
        mu = np.ascontiguousarray(mu, dtype=np.float64).reshape(-1)
        if mu.size != self.n_rows:
            raise ValueError("mu has %d entries, expected %d" % (mu.size, self.n_rows))
        vals = np.empty(self.hess_nnz) if vals is None else vals
        self._chk(self._L.pcl_hess(self._h, _ptr(Z), _ptr(mu), _ptr(vals)))
        return vals

    # -- device-pointer calls (torch tensors on this context's GPU; asynchronous) ----------------
    def set_stream(self, stream_handle):
        """Launch on this hipStream_t handle (0 = the legacy default stream); ``None`` restores
        the context's own stream."""
        if stream_handle is None:
            self._chk(self._L.pcl_reset_stream(self._h))
        else:
            self._chk(self._L.pcl_set_stream(self._h, ctypes.c_void_p(int(stream_handle))))

    def sync(self):
        self._chk(self._L.pcl_sync(self._h))

    def eval_dev(self, Z, delta):
        self._chk(self._L.pcl_eval_dev(self._h, _ptr(Z), _ptr(delta)))

    def eval_jac_dev(self, Z, delta, vals):
        self._chk(self._L.pcl_eval_jac_dev(self._h, _ptr(Z), _ptr(delta), _ptr(vals)))

    def eval_jac_compact_dev(self, Z, delta, compact):
        self._chk(self._L.pcl_eval_jac_compact_dev(self._h, _ptr(Z), _ptr(delta), _ptr(compact)))

    def jac_expand_dev(self, compact, vals):
        self._chk(self._L.pcl_jac_expand_dev(self._h, _ptr(compact), _ptr(vals)))

    def hess_dev(self, Z, mu, vals):
        self._chk(self._L.pcl_hess_dev(self._h, _ptr(Z), _ptr(mu), _ptr(vals)))

    def jac_dev(self, Z, vals):
        self._chk(self._L.pcl_jac_dev(self._h, _ptr(Z), _ptr(vals)))

    # -- member window: the evaluator calls cover members [first, first+count) (one member of the reference's integrator vector)
    def set_member_window(self, first=0, count=None):
        count = self.batch - first if count is None else count
        self._chk(self._L.pcl_set_member_window(self._h, int(first), int(count)))
        self.window = (int(first), int(count))
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._chk(self._L.pcl_constraint_dim(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        self.n_rows = b.value
        self._chk(self._L.pcl_jac_nnz(self._h, ctypes.byref(a), ctypes.byref(b)))
        self.jac_nnz = a.value
        self._chk(self._L.pcl_hess_nnz(self._h, ctypes.byref(a), ctypes.byref(b)))
        self.hess_nnz = a.value
        self._chk(self._L.pcl_jac_compact_nnz(self._h, ctypes.byref(a), ctypes.byref(b)))
        self.compact_nnz = a.value

    # -- DerivativeIntegrator / time-consistency rows on this context's trajectory layout -----------------
    def deriv_dims(self, dx_off, dim):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._chk(self._L.pcl_deriv_nnz(self._h, dx_off, dim, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def deriv_structure(self, x_off, dx_off, dim):
        _, nnz = self.deriv_dims(dx_off, dim)
        rows, cols = np.empty(nnz, np.int64), np.empty(nnz, np.int64)
        p = ctypes.POINTER(ctypes.c_int64)
        self._chk(self._L.pcl_deriv_structure(self._h, x_off, dx_off, dim, rows.ctypes.data_as(p), cols.ctypes.data_as(p)))
        return rows, cols

    def deriv_eval_jac(self, x_off, dx_off, dim, Z):
        Z = self._z(Z)
        nr, nnz = self.deriv_dims(dx_off, dim)
        delta, vals = np.empty(nr), np.empty(nnz)
        self._chk(self._L.pcl_deriv_eval_jac(self._h, x_off, dx_off, dim, _ptr(Z), _ptr(delta), _ptr(vals)))
        return delta, vals

    def deriv_eval_jac_dev(self, x_off, dx_off, dim, Z, delta, vals):
        self._chk(self._L.pcl_deriv_eval_jac_dev(self._h, x_off, dx_off, dim, _ptr(Z), _ptr(delta), _ptr(vals)))

    # -- terminal infidelity objective on device ----------------------------------------------------------------
    def set_goal(self, goal_iso_vec):
        g = np.ascontiguousarray(goal_iso_vec, dtype=np.float64).reshape(-1)
        if g.size != self.x_dim:
            raise ValueError("goal iso-vec has %d entries, expected %d" % (g.size, self.x_dim))
        self._chk(self._L.pcl_set_goal(self._h, _ptr(g)))

    def set_goal_subspace(self, goal_sub_iso_vec, subspace):
        """Embedded goal: iso-vec of the ns x ns block ``unembed(op)`` and its 0-based subspace indices."""
        sub = np.ascontiguousarray(subspace, dtype=np.int32).reshape(-1)
        g = np.ascontiguousarray(goal_sub_iso_vec, dtype=np.float64).reshape(-1)
        if g.size != 2 * sub.size * sub.size:
            raise ValueError("subspace goal iso-vec has %d entries, expected %d" % (g.size, 2 * sub.size * sub.size))
        self._chk(self._L.pcl_set_goal_subspace(self._h, _ptr(g), sub.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), sub.size))

    def set_goal_form(self, scope, A, c):
        """Terminal loss ``Q w |1 - F(x)|``, ``F = c'x + sum_r (A_r'x)^2`` (``pcl_set_goal_form``): scope 0 per member, 1 joint."""
        L = self.x_dim * (self.batch if scope else 1)
        A = None if A is None else np.ascontiguousarray(A, dtype=np.float64).reshape(-1, L)
        c = None if c is None else np.ascontiguousarray(c, dtype=np.float64).reshape(L)
        self._L.pcl_set_goal_form.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self._L.pcl_set_goal_form(self._h, int(scope), 0 if A is None else A.shape[0], None if A is None else A.ctypes.data, None if c is None else c.ctypes.data))

    def objective_hess_structure(self):
        n = ctypes.c_int64()
        self._chk(self._L.pcl_objective_hess_nnz(self._h, ctypes.byref(n)))
        rows, cols = np.empty(n.value, dtype=np.int64), np.empty(n.value, dtype=np.int64)
        self._L.pcl_objective_hess_structure.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(self._L.pcl_objective_hess_structure(self._h, rows.ctypes.data, cols.ctypes.data))
        return rows, cols

    def objective_hess(self, Z, Q, sigma=1.0):
        Z = self._z(Z)
        n = ctypes.c_int64()
        self._chk(self._L.pcl_objective_hess_nnz(self._h, ctypes.byref(n)))
        vals = np.empty(n.value)
        self._L.pcl_objective_hess.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        self._chk(self._L.pcl_objective_hess(self._h, Z.ctypes.data, float(Q), float(sigma), vals.ctypes.data))
        return vals

    def objective_hess_dev(self, Z, Q, sigma, vals):
        self._L.pcl_objective_hess_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        self._chk(self._L.pcl_objective_hess_dev(self._h, _ptr(Z), float(Q), float(sigma), _ptr(vals)))

    def set_weights(self, weights):
        if weights is None:
            self._chk(self._L.pcl_set_weights(self._h, None))
            return
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
        if w.size != self.batch:
            raise ValueError("expected %d weights, got %d" % (self.batch, w.size))
        self._chk(self._L.pcl_set_weights(self._h, _ptr(w)))

    def infidelity_dev(self, Z, Q, value=None, grad=None):
        self._chk(self._L.pcl_infidelity_dev(self._h, _ptr(Z), float(Q), _ptr(value), _ptr(grad)))

    def add_regularizer(self, off, dim, R, dt_power=2):
        R = np.ascontiguousarray(np.broadcast_to(np.asarray(R, dtype=np.float64), (dim,)))
        self._chk(self._L.pcl_add_regularizer(self._h, int(off), int(dim), _ptr(R), int(dt_power)))

    def clear_regularizers(self):
        self._chk(self._L.pcl_clear_regularizers(self._h))

    def objective_dev(self, Z, Q, value, grad=None):
        self._chk(self._L.pcl_objective_dev(self._h, _ptr(Z), float(Q), _ptr(value), _ptr(grad)))

    def objective(self, Z, Q, want_grad=True):
        """(value [1 or batch], gradient [z_len] or None) of the whole objective, host buffers."""
        Z = self._z(Z)
        value = np.empty(self.batch if self.batch_mode == PCL_BATCH_TRAJ else 1)
        grad = np.empty(self.z_len) if want_grad else None
        self._chk(self._L.pcl_objective(self._h, _ptr(Z), float(Q), _ptr(value), _ptr(grad)))
        return value, grad

    def merit_grad_len(self):
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._chk(self._L.pcl_merit_grad_len(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def merit_grad_dev(self, delta, lam, vals, out):
        """out <- [phi | J^T lam on the shared u_k | ... on dt_k] (lam None: lam = delta, phi = the constraint merit)."""
        self._chk(self._L.pcl_merit_grad_dev(self._h, _ptr(delta), _ptr(lam), _ptr(vals), _ptr(out)))

    def eval_jac_merit_dev(self, Z, lam, delta, vals, out):
        """``eval_jac_dev`` + ``merit_grad_dev`` in one pass over the state columns (pcl_eval_jac_merit_dev): the fused kernel
        forms the payload's dot products while a column's vectors are in LDS; same outputs as the two calls."""
        self._chk(self._L.pcl_eval_jac_merit_dev(self._h, _ptr(Z), _ptr(lam), _ptr(delta), _ptr(vals), _ptr(out)))

    def eval_jac_merit_objective_dev(self, Z, lam, delta, vals, out, Q, value, grad):
        """``objective_dev`` + ``eval_jac_merit_dev`` -- a rank's whole step of a sharded ensemble -- in two launches instead of four
        (pcl_eval_jac_merit_objective_dev); the same bits."""
        self._chk(self._L.pcl_eval_jac_merit_objective_dev(self._h, _ptr(Z), _ptr(lam), _ptr(delta), _ptr(vals), _ptr(out), float(Q), _ptr(value), _ptr(grad)))

    # -- rollout (exact piecewise-constant propagation from the knot-0 state) ----------------------------------------
    def rollout(self, Z, out=None):
        """[batch, N, x_dim] iso-vec states: X_{k+1} = exp(dt_k G(u_k)) X_k."""
        Z = self._z(Z)
        out = np.empty((self.window[1], self.N, self.x_dim)) if out is None else out
        self._chk(self._L.pcl_rollout(self._h, _ptr(Z), _ptr(out)))
        return out

    def rollout_dev(self, Z, out):
        self._chk(self._L.pcl_rollout_dev(self._h, _ptr(Z), _ptr(out)))

    # -- RCCL (C-ABI path; torch.distributed is the alternative plumbing, see distributed.py) ----------------
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        rc = self._L.pcl_comm_get_unique_id(buf)
        if rc != 0:
            raise PclError(rc, (self._L.pcl_last_error(None) or b"").decode())
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        self._chk(self._L.pcl_comm_init(self._h, ctypes.create_string_buffer(unique_id, 128), rank, nranks))

    def reduce_sum_dev(self, buf):
        self._chk(self._L.pcl_reduce_sum_dev(self._h, _ptr(buf), buf.numel()))

    def reduce_sum(self, buf):
        """In-place sum over the ranks of a host (numpy, float64, contiguous) buffer."""
        self._chk(self._L.pcl_reduce_sum(self._h, _ptr(buf), buf.size))
        return buf

    def set_option(self, key, value):
        self._chk(self._L.pcl_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = ctypes.c_int64()
        self._chk(self._L.pcl_get_option(self._h, key.encode(), ctypes.byref(v)))
        return v.value


def _abs_bound(b, n):
    """max(|lower|, |upper|) per component of a bounds entry, in the shapes the mirror and the reference use: ``(lower_vec, upper_vec)`` (NamedTrajectory
    [REF named_trajectory_conversion.jl:331-332]), a list of ``(lo, hi)`` pairs per component (``system.drive_bounds``), one ``(lo, hi)`` pair, a symmetric
    bound per component, or a scalar."""
    if isinstance(b, tuple) and len(b) == 2 and all(np.ndim(x) <= 1 for x in b):  # (lower, upper): vectors or scalars
        lo, hi = (np.broadcast_to(np.asarray(x, dtype=np.float64), (n,)) for x in b)
        return np.maximum(np.abs(lo), np.abs(hi))
    a = np.asarray(b, dtype=np.float64)
    if a.ndim == 0:
        return np.full(n, abs(float(a)))
    if a.shape == (n, 2):  # pairs per component
        return np.max(np.abs(a), axis=1)
    if a.shape == (2, n):
        return np.max(np.abs(a), axis=0)
    if a.shape == (n,):
        return np.abs(a)
    if a.shape == (2,):
        return np.full(n, np.max(np.abs(a)))
    raise ValueError("bounds of shape %r for a component of dimension %d" % (a.shape, n))


def _decide_order(ctx, traj, u_name, m, tol, copies=1):
    """``pade_order = 0`` at construction: the order policy over the trajectory's bounds, else over the trajectory itself."""
    ub, tb = traj.bounds.get(u_name), traj.bounds.get(traj.timestep)
    if ub is not None and tb is not None and m:
        umax = _abs_bound(ub, len(traj.components[u_name]))[:m]
        return ctx.set_order_policy(float(np.max(_abs_bound(tb, len(traj.components[traj.timestep])))), umax, tol)
    Z = np.ascontiguousarray(traj.datavec, dtype=np.float64).reshape(-1)
    return ctx.set_order_from_trajectory(np.tile(Z, copies) if copies > 1 else Z, tol)


def _resolved_order(ctx):
    """The order a secondary context (the scalar form ``f``) must be created with: the main context's, never 0."""
    order = ctx.pade_order
    if order == 0:
        raise PclError(-1, "the Pade order of this integrator is not decided yet (pade_order = 0 and no policy / trajectory seen)")
    return order


class HipPadeIntegrator:
    """Drop-in for ``DirectTrajOpt.BilinearIntegrator`` on the unitary path.

    One integrator object covers one state component (``x_name``) or, for a
    ``SamplingTrajectory``, the M member components at once (the reference builds
    a ``Vector{BilinearIntegrator}``, one per member, whose rows are concatenated
    in order -- the row order here is identical: member-major).
    """

    def __init__(self, G_drift, G_drives, traj, x_name=STATE, u_name="u", *, device=0, index_base=0, pade_order=0, order_tol=1e-10):
        """``pade_order=0`` (the default): the smallest diagonal Pade order whose deviation from the reference's exp constraint
        [REF docs/src/concepts/index.md:21] stays below ``order_tol`` over the trajectory's bounds on ``u`` and the timestep
        (``traj.bounds``); without bounds, over ``traj`` itself (x 1.5).  Decided HERE, so every entry point -- host or device
        pointers, the scalar form ``f`` -- evaluates one order from the first call on.  ``pade_order=2..10`` pins the order
        (BASELINE.json's metric is quoted on 4, which deviates from the exp constraint by 1.6e-5 at config 3)."""
        x_names = [x_name] if isinstance(x_name, str) else list(x_name)
        G_drives = np.asarray(G_drives, dtype=np.float64)
        G_drift = np.asarray(G_drift, dtype=np.float64)
        per_member = G_drift.ndim == 3
        n = G_drift.shape[-1]
        d = n // 2
        m = G_drives.shape[0] if G_drives.size else 0
        if per_member and G_drift.shape[0] != len(x_names):
            raise ValueError("one G_drift per member expected (%d != %d)" % (G_drift.shape[0], len(x_names)))
        for nm in x_names:
            if nm not in traj.components:
                raise KeyError("trajectory has no component %r" % (nm,))
        xlen = len(traj.components[x_names[0]])
        vec = n % 2 == 1  # odd generator dimension: a compact density vector (levels^2) under a compact Lindbladian
        if vec:
            if xlen != n or any(len(traj.components[nm]) != xlen for nm in x_names):
                raise ValueError("an odd generator dimension (%d) takes state components of that length; got %d" % (n, xlen))
            d, cols = n, 1
        else:
            if xlen % n or not (1 <= xlen // n <= d) or any(len(traj.components[nm]) != xlen for nm in x_names):
                raise ValueError("state components must all have dim n*C with n = %d, 1 <= C <= d (unitary: C = d = %d, ket: C = 1); got %d"
                                 % (n, d, xlen))  # fmt: skip
            cols = xlen // n
        if m and len(traj.components[u_name]) < m:
            raise ValueError("drive component %r has dim %d < n_drives = %d" % (u_name, len(traj.components[u_name]), m))
        self.x_names = x_names
        self.x_name = x_names[0] if len(x_names) == 1 else tuple(x_names)
        self.u_name = u_name
        self.G_drift, self.G_drives = G_drift, G_drives.reshape(m, n, n)
        self._sig = (traj.dim, traj.N)
        self._ctx = _PclContext(
            d=d, m=m, N=traj.N, z_dim=traj.dim, u_off=traj.components[u_name].start,
            dt_off=traj.components[traj.timestep].start, x_offs=[traj.components[nm].start for nm in x_names],
            G0=G_drift, Gj=self.G_drives, batch=len(x_names), batch_mode=PCL_BATCH_MEMBERS, per_member_G0=per_member,
            global_dim=traj.global_dim, device=device, index_base=index_base, pade_order=pade_order,
            state_cols=_lib.PCL_STATE_VECTOR if vec else cols,
        )  # fmt: skip
        if pade_order == 0:
            _decide_order(self._ctx, traj, u_name, m, order_tol)
        self._state_cols = _lib.PCL_STATE_VECTOR if vec else cols
        self.x_dim = self._ctx.x_dim * len(x_names) if len(x_names) > 1 else self._ctx.x_dim
        self.dim = self._ctx.n_rows
        self._f_ctx = None

    # ---- reference-style properties ------------------------------------------------------------
    @property
    def ctx(self):
        return self._ctx

    @property
    def pade_order(self):
        return self._ctx.pade_order

    def _check(self, traj):
        if (traj.dim, traj.N) != self._sig:
            raise ValueError("trajectory shape (dim=%d, N=%d) differs from the one this integrator was built for %r"
                             % (traj.dim, traj.N, self._sig))  # fmt: skip

    def f(self, x_next, x, u, dt):
        """Scalar (one-interval) form ``B.f(x_next, x, u, dt)`` [REF integrators.jl:525]."""
        c = self._ctx
        if len(self.x_names) != 1:
            raise NotImplementedError("f is defined for single-state integrators")
        if self._f_ctx is None:
            self._f_ctx = _PclContext(d=c.d, m=c.m, N=2, z_dim=c.x_dim + 1 + c.m, u_off=c.x_dim + 1, dt_off=c.x_dim,
                                      x_offs=[0], G0=self.G_drift, Gj=self.G_drives, batch=1,
                                      batch_mode=PCL_BATCH_MEMBERS, state_cols=self._state_cols, pade_order=_resolved_order(c))  # fmt: skip
        z = np.zeros((2, c.x_dim + 1 + c.m))
        z[0, : c.x_dim], z[0, c.x_dim], z[0, c.x_dim + 1 :] = x, dt, np.asarray(u)[: c.m]
        z[1, : c.x_dim] = x_next
        return self._f_ctx.eval(z)

    def close(self):
        self._ctx.close()
        if self._f_ctx is not None:
            self._f_ctx.close()


class _EnsembleCore:
    """What the M per-member integrators of a SamplingTrajectory share: ONE batched ``pcl_ctx`` (all members, per-member
    drift tiles, one launch) and the results of the last fused evaluation.

    The reference evaluates its integrators one after the other on the same trajectory
    (``evaluate!(delta_i, B_i, traj)`` for i = 1..M [REF src/control/integrators.jl:316-317]); here the first member asked
    about a trajectory launches the fused kernel for ALL members and the others slice their rows out of the cached
    result.  The cache key is the trajectory buffer itself (byte comparison with the copy the results were computed from),
    so a stale result can never be served.  Hessian calls carry a per-member multiplier slice and run on a one-member
    window of the same context (``pcl_set_member_window``)."""

    def __init__(self, fused, n_members):
        self.fused = fused  # HipPadeIntegrator over all members
        self.ctx = fused.ctx
        self.M = n_members
        self.per_rows = self.ctx.x_dim * self.ctx.K
        self.per_jac = self.ctx.jac_per * self.ctx.K
        self.per_hess = self.ctx.hess_per * self.ctx.K
        self._Z = None
        self._delta = None
        self._vals = None
        self.launches = 0  # fused launches so far (tests check the sharing)

    def _fresh(self, Z):
        Z = np.ascontiguousarray(Z, dtype=np.float64).reshape(-1)
        if self._Z is None or self._Z.shape != Z.shape or not np.array_equal(self._Z, Z):
            self._Z = Z.copy()
            self._delta = self._vals = None
        return self._Z

    def delta(self, Z):
        Z = self._fresh(Z)
        if self._delta is None:
            self.ctx.set_member_window(0, self.M)
            self._delta = self.ctx.eval(Z)
            self.launches += 1
        return self._delta

    def delta_and_vals(self, Z):
        Z = self._fresh(Z)
        if self._vals is None:
            self.ctx.set_member_window(0, self.M)
            self._delta, self._vals = self.ctx.eval_jac(Z)
            self.launches += 1
        return self._delta, self._vals

    def close(self):
        self.fused.close()


class _MemberView:
    """The ``ctx``-like object of one member integrator: the generic functions (``evaluate_``, ``eval_jacobian`` ...) call
    ``eval / jac / hess / *_structure`` on it exactly as on a whole context; rows are numbered inside the member's block
    (the reference's integrators are separate objects; the problem adds each block's row offset)."""

    def __init__(self, core, i):
        self._core, self._i = core, i
        c = core.ctx
        self.x_dim, self.K, self.N, self.z_dim, self.z_len = c.x_dim, c.K, c.N, c.z_dim, c.z_len
        self.n_rows, self.jac_nnz, self.hess_nnz = core.per_rows, core.per_jac, core.per_hess

    def eval(self, Z, delta=None):
        i, r = self._i, self._core.per_rows
        d = self._core.delta(Z)[i * r : (i + 1) * r]
        if delta is not None:
            delta[:] = d
            return delta
        return d.copy()

    def jac(self, Z, vals=None):
        i, r = self._i, self._core.per_jac
        v = self._core.delta_and_vals(Z)[1][i * r : (i + 1) * r]
        if vals is not None:
            vals[:] = v
            return vals
        return v.copy()

    def eval_jac(self, Z, delta=None, vals=None):
        return self.eval(Z, delta), self.jac(Z, vals)

    def _windowed(self, fn):
        c = self._core.ctx
        c.set_member_window(self._i, 1)
        try:
            return fn(c)
        finally:
            c.set_member_window(0, self._core.M)

    def jac_structure(self, dtype=np.int64):
        return self._windowed(lambda c: c.jac_structure(dtype))

    def hess_structure(self, dtype=np.int64):
        return self._windowed(lambda c: c.hess_structure(dtype))

    def hess(self, Z, mu, vals=None):
        return self._windowed(lambda c: c.hess(Z, mu, vals))

    def rollout(self, Z):
        return self._windowed(lambda c: c.rollout(Z))


class HipPadeMemberIntegrator:
    """One element of ``BilinearIntegrator(qtraj::SamplingTrajectory, N)``'s ``Vector{BilinearIntegrator}``
    [REF src/control/integrators.jl:134-162]: the dynamics rows of ONE ensemble member (state component ``x_name``, the
    member's own system), with the reference's properties ``dim == x_dim*(N-1)``, ``x_dim``, ``x_name``, ``f``.  All members
    of one vector evaluate through a shared batched context (``ensemble``)."""

    def __init__(self, core, i, x_name, G_drift, G_drives, u_name, sig):
        self.ensemble, self.member = core, i
        self.x_name, self.x_names, self.u_name = x_name, [x_name], u_name
        self.G_drift, self.G_drives = G_drift, G_drives
        self._sig = sig
        self._view = _MemberView(core, i)
        self.x_dim = core.ctx.x_dim
        self.dim = core.per_rows
        self._f_ctx = None

    @property
    def ctx(self):
        return self._view

    @property
    def pade_order(self):  # (read from the shared context: never a stale copy of the constructor's argument)
        return self.ensemble.ctx.pade_order

    _check = HipPadeIntegrator._check

    def f(self, x_next, x, u, dt):
        """``B.f(x_next, x, u, dt)`` of this member's system [REF integrators.jl:518-525]."""
        c = self.ensemble.ctx
        if self._f_ctx is None:
            self._f_ctx = _PclContext(d=c.d, m=c.m, N=2, z_dim=c.x_dim + 1 + c.m, u_off=c.x_dim + 1, dt_off=c.x_dim, x_offs=[0],
                                      G0=self.G_drift, Gj=self.G_drives, batch=1, batch_mode=PCL_BATCH_MEMBERS,
                                      state_cols=self.ensemble.fused._state_cols, pade_order=_resolved_order(c))  # fmt: skip
        z = np.zeros((2, c.x_dim + 1 + c.m))
        z[0, : c.x_dim], z[0, c.x_dim], z[0, c.x_dim + 1 :] = x, dt, np.asarray(u)[: c.m]
        z[1, : c.x_dim] = x_next
        return self._f_ctx.eval(z)

    def close(self):
        if self._f_ctx is not None:
            self._f_ctx.close()
            self._f_ctx = None
        if self.member == 0:
            self.ensemble.close()


class HipPadeMultistart:
    """B independent trajectories of identical shape evaluated in one launch (multistart
    seeds; BASELINE.json config 5).  Not a reference type: the reference has no multistart
    facility; each seed is its own NLP and owns rows/columns ``b``-major."""

    def __init__(self, G_drift, G_drives, traj, batch, x_name=STATE, u_name="u", *, device=0, index_base=0, pade_order=0, order_tol=1e-10):
        G_drives = np.asarray(G_drives, dtype=np.float64)
        n = np.asarray(G_drift).shape[-1]
        m = G_drives.shape[0] if G_drives.size else 0
        xlen = len(traj.components[x_name])
        if xlen % n or not (1 <= xlen // n <= n // 2):
            raise ValueError("state component %r has dim %d, expected n*C with n = %d" % (x_name, xlen, n))
        self._ctx = _PclContext(
            d=n // 2, m=m, N=traj.N, z_dim=traj.dim, u_off=traj.components[u_name].start,
            dt_off=traj.components[traj.timestep].start, x_offs=[traj.components[x_name].start], G0=G_drift,
            Gj=G_drives.reshape(m, n, n), batch=batch, batch_mode=PCL_BATCH_TRAJ, device=device, index_base=index_base,
            state_cols=xlen // n, pade_order=pade_order,
        )  # fmt: skip
        if pade_order == 0:  # (decided from the bounds, else from `traj` -- the shape-defining seed -- as HipPadeIntegrator does)
            _decide_order(self._ctx, traj, u_name, m, order_tol, batch)
        self.batch = batch
        self.x_name = x_name
        self.x_dim = self._ctx.x_dim
        self.dim = self._ctx.n_rows

    @property
    def ctx(self):
        return self._ctx

    def close(self):
        self._ctx.close()


class DerivativeIntegrator:
    """``DerivativeIntegrator(x, dx, traj)``: rows ``x_{k+1} - x_k - dt_k dx_k`` of the problem templates' integrator
    list ``[dynamics, DerivativeIntegrator(u, du), DerivativeIntegrator(du, ddu)]``
    [REF src/control/templates/smooth_pulse_problem.jl:264-275].  ``dx_name=None`` gives the time-consistency rows
    ``t_{k+1} - t_k - dt_k`` [REF :277].  Evaluated on the GPU through the dynamics integrator's context (same
    trajectory layout, same stream)."""

    def __init__(self, x_name, dx_name, traj, like):
        self._ctx = like.ctx
        self._sig = (traj.dim, traj.N)
        self.x_name, self.dx_name = x_name, dx_name
        self.x_off = traj.components[x_name].start
        self.dx_off = -1 if dx_name is None else traj.components[dx_name].start
        self.x_dim = len(traj.components[x_name])
        if dx_name is not None and len(traj.components[dx_name]) != self.x_dim:
            raise ValueError("components %r and %r differ in length" % (x_name, dx_name))
        self.dim, self.nnz = self._ctx.deriv_dims(self.dx_off, self.x_dim)

    @property
    def ctx(self):
        return self

    def _check(self, traj):
        if (traj.dim, traj.N) != self._sig:
            raise ValueError("trajectory shape differs from the one this integrator was built for")

    # the generic functions below dispatch on these three
    def eval(self, Z, delta=None):
        d, _ = self._ctx.deriv_eval_jac(self.x_off, self.dx_off, self.x_dim, Z)
        if delta is not None:
            delta[:] = d
            return delta
        return d

    def jac(self, Z):
        return self._ctx.deriv_eval_jac(self.x_off, self.dx_off, self.x_dim, Z)[1]

    def jac_structure(self, dtype=np.int64):
        r, c = self._ctx.deriv_structure(self.x_off, self.dx_off, self.x_dim)
        return r.astype(dtype), c.astype(dtype)


# ---------------------------------------------------------------------------
# reference-style generic functions
# ---------------------------------------------------------------------------
def BilinearIntegrator(system, traj, x_name=None, u_name="u", **kw):
    """``BilinearIntegrator(qtraj, N)`` for the time-independent unitary path.

    ``system`` is one system (UnitaryTrajectory) -> one integrator, or a list of systems
    (``BilinearIntegrator(qtraj::SamplingTrajectory, N)`` [REF integrators.jl:134-146]) -> a LIST with one integrator per
    member, in member order, as the reference returns and as ``SamplingProblem`` requires
    [REF src/control/templates/sampling_problem.jl:190-223].  Members that share the drive generators (per-member
    ``H_drift`` only -- BASELINE config 4) evaluate through one batched context; an ensemble whose members differ in
    their drive generators too (each member uses its full ``sys.G`` [REF integrators.jl:149-162]) gets one context per
    member."""
    if isinstance(system, (list, tuple)):
        from .quantum import OpenQuantumSystem
        from .trajectory import DENSITY

        systems = list(system)
        if any(getattr(s, "time_dependent", False) for s in systems):
            raise NotImplementedError("time-dependent systems use TimeDependentBilinearIntegrator (out of scope)")
        # Every base the reference samples [REF integrators.jl:149-226 (_sampling_integrator)]: unitary and ket members carry one
        # state each, MultiKet / MultiDensity members a LIST of sub-states (one integrator per sub-state, all on the member's
        # system), density members their compact Lindbladian generators.  `x_name`: one entry per member, a name or a list of
        # sub-state names; the result is flat, member-major, sub-states in order -- what the reference's reduce(vcat, ...) returns.
        default = DENSITY if isinstance(systems[0], OpenQuantumSystem) else STATE
        names = x_name or ["%s%d" % (default, i) for i in range(1, len(systems) + 1)]
        if len(names) != len(systems):
            raise ValueError("%d state names for %d systems" % (len(names), len(systems)))
        per = [[nm] if isinstance(nm, str) else list(nm) for nm in names]
        flat = [nm for p_ in per for nm in p_]
        owner = [i for i, p_ in enumerate(per) for _ in p_]
        Gd = [s.G_drives_array() for s in systems]

        def fused_members(idx):  # one batched context over the (member, sub-state) pairs `idx` that share drive generators
            fused = HipPadeIntegrator(np.array([systems[owner[j]].G_drift for j in idx]), Gd[owner[idx[0]]], traj, [flat[j] for j in idx], u_name, **kw)
            core = _EnsembleCore(fused, len(idx))
            return [HipPadeMemberIntegrator(core, a, flat[j], systems[owner[j]].G_drift, fused.G_drives, u_name, fused._sig)
                    for a, j in enumerate(idx)]  # fmt: skip

        if all(np.array_equal(Gd[0], g) for g in Gd[1:]):
            return fused_members(list(range(len(flat))))
        out = []  # members that differ in their drive generators too: a context per member (shared by its sub-states)
        for i, s in enumerate(systems):
            idx = [j for j in range(len(flat)) if owner[j] == i]
            out += [HipPadeIntegrator(s.G_drift, Gd[i], traj, flat[idx[0]], u_name, **kw)] if len(idx) == 1 else fused_members(idx)
        return out
    if getattr(system, "time_dependent", False):
        raise NotImplementedError("time-dependent systems use TimeDependentBilinearIntegrator (out of scope)")
    from .quantum import OpenQuantumSystem
    from .trajectory import DENSITY

    if isinstance(system, OpenQuantumSystem):  # BilinearIntegrator(qtraj::DensityTrajectory, N) [REF integrators.jl:82-95]
        return HipPadeIntegrator(system.G_drift, system.G_drives_array(), traj, x_name or DENSITY, u_name, **kw)
    return HipPadeIntegrator(system.G_drift, system.G_drives_array(), traj, x_name or STATE, u_name, **kw)


def evaluate_(delta, B, traj):
    """``evaluate!(delta, B, traj)``: fills ``delta`` (length ``B.dim``) in place."""
    B._check(traj)
    if delta.shape != (B.dim,):
        raise ValueError("delta must have length B.dim = %d" % B.dim)
    B.ctx.eval(traj.datavec, delta)
    return delta


def jacobian_structure(B, dtype=np.int64):
    return B.ctx.jac_structure(dtype)


def hessian_structure(B, dtype=np.int64):
    return B.ctx.hess_structure(dtype)


def eval_jacobian(B, traj):
    """``eval_jacobian(B, traj)`` -> scipy.sparse CSR of shape
    ``(B.dim, traj.dim*traj.N + traj.global_dim)``."""
    import scipy.sparse as sp

    B._check(traj)
    vals = B.ctx.jac(traj.datavec)
    rows, cols = B.ctx.jac_structure()
    return sp.csr_matrix((vals, (rows, cols)), shape=(B.dim, traj.dim * traj.N + traj.global_dim))


def eval_hessian_of_lagrangian(B, traj, mu):
    """Symmetric ``sum_k mu_k^T grad^2 delta_k`` as scipy.sparse CSR (full, both triangles)."""
    import scipy.sparse as sp

    B._check(traj)
    vals = B.ctx.hess(traj.datavec, mu)
    rows, cols = B.ctx.hess_structure()
    nv = traj.dim * traj.N + traj.global_dim
    L = sp.coo_matrix((vals, (rows, cols)), shape=(nv, nv)).tocsr()
    return L + sp.tril(L, -1).T


def unitary_rollout(B, traj):
    """``unitary_rollout(traj, sys; interpolation = :constant)`` [REF src/quantum/dynamics.jl:631-667]: the iso-vec
    states at every knot, ``x_dim x N`` (a list of such arrays for an ensemble integrator), by exact propagation
    ``exp(dt_k G(u_k))`` on the GPU instead of an ODE solve.  Starts from the trajectory's knot-0 state."""
    B._check(traj)
    X = B.ctx.rollout(traj.datavec)
    outs = [X[i].T.copy() for i in range(X.shape[0])]
    return outs[0] if len(outs) == 1 else outs


def unitary_rollout_fidelity(B, traj, U_goal=None):
    """``unitary_rollout_fidelity`` [REF src/quantum/dynamics.jl:594-629]: ``|tr(U_goal' U_N)|^2 / d^2`` of the rolled-out
    terminal state(s); ``U_goal`` defaults to the trajectory's goal for the integrator's state."""
    from .quantum import iso_vec_to_operator

    X = unitary_rollout(B, traj)
    Xs = X if isinstance(X, list) else [X]
    fids = []
    for name, Xi in zip(B.x_names, Xs):
        Ug = np.asarray(U_goal) if U_goal is not None else iso_vec_to_operator(traj.goal[name])
        Uf = iso_vec_to_operator(Xi[:, -1])
        fids.append(abs(np.trace(Ug.conj().T @ Uf)) ** 2 / Ug.shape[0] ** 2)
    return fids[0] if len(fids) == 1 else fids

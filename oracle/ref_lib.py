"""ctypes loader for oracle/_build/libpade_ref.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/pade_ref.c.  Used by tests/, smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpade_ref.so")
_SO_EXPV = os.path.join(_HERE, "_build", "libexpv_ref.so")  # the reference's ALGORITHM restated (expv + forward-mode duals): expv_ref.c
_lib = None
_lib_expv = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pade_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "_build/libpade_ref.so"])
    if force or not os.path.exists(_SO_EXPV) or os.path.getmtime(_SO_EXPV) < os.path.getmtime(os.path.join(_HERE, "expv_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "_build/libexpv_ref.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.pade_ref_eval_jac.argtypes = [ctypes.c_int] * 7 + [dp, dp, dp, dp, dp, ctypes.c_int]
        _lib.pade_ref_hess.argtypes = [ctypes.c_int] * 7 + [dp, dp, dp, dp, dp, ctypes.c_int]
        _lib.pade_ref_jac_nnz_per_interval.restype = ctypes.c_long
        _lib.pade_ref_hess_nnz_per_interval.restype = ctypes.c_long
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _colmajor(G0, Gj):
    """numpy [i,j] matrices -> flat column-major buffers."""
    g0 = np.ascontiguousarray(np.asarray(G0, dtype=np.float64).T)
    gj = np.ascontiguousarray(np.transpose(np.asarray(Gj, dtype=np.float64), (0, 2, 1))) if len(Gj) else np.zeros(1)
    return g0, gj


def eval_jac(Z, lay, G0, Gj, x_off=None, want_jac=True, nthreads=0, out=None):
    """Z: [N, z_dim] C-contiguous (== z_dim x N column-major).  Returns (delta [K,x_dim], jac [K,per] | None).
    ``out=(delta, jac)`` reuses preallocated outputs (timing runs)."""
    L = lib()
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    g0, gj = _colmajor(G0, Gj)
    o = lay.x_off if x_off is None else x_off
    per = L.pade_ref_jac_nnz_per_interval(lay.d, lay.m)
    if out is not None:
        delta, jac = out
    else:
        delta = np.empty((lay.K, lay.x_dim))
        jac = np.empty((lay.K, per)) if want_jac else None
    rc = L.pade_ref_eval_jac(lay.d, lay.m, lay.N, lay.z_dim, o, lay.u_off, lay.dt_off, _p(g0), _p(gj), _p(Z), _p(delta), _p(jac), nthreads)
    assert rc == 0
    return delta, jac


def hess(Z, mu, lay, G0, Gj, x_off=None, nthreads=0):
    L = lib()
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    g0, gj = _colmajor(G0, Gj)
    o = lay.x_off if x_off is None else x_off
    per = L.pade_ref_hess_nnz_per_interval(lay.d, lay.m)
    out = np.empty((lay.K, per))
    rc = L.pade_ref_hess(lay.d, lay.m, lay.N, lay.z_dim, o, lay.u_off, lay.dt_off, _p(g0), _p(gj), _p(Z), _p(mu), _p(out), nthreads)
    assert rc == 0
    return out


def expv_eval_jac(Z, lay, G0, Gj, x_off=None, nthreads=0, k_first=0, k_count=None, chunk=12, want_jac=True):
    """oracle/expv_ref.c: delta_k = x_{k+1} - expv(dt_k, Ghat(u_k), x_k) and its Jacobian by forward-mode duals through expv in chunks of `chunk`
    directions (the reference's algorithm, restated; CPU baseline only).  Intervals k_first .. k_first + k_count - 1; returns
    (delta [K, x_dim], jac [K, per] | None, {"off_structure_max", "taylor_terms"}) -- rows of intervals outside the range are NaN."""
    global _lib_expv
    if _lib_expv is None:
        if not os.path.exists(_SO_EXPV):
            build()
        _lib_expv = ctypes.CDLL(_SO_EXPV)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib_expv.expv_ref_eval_jac.argtypes = [ctypes.c_int] * 7 + [dp, dp, dp, dp, dp] + [ctypes.c_int] * 4 + [dp, ctypes.POINTER(ctypes.c_long)]
        _lib_expv.expv_ref_jac_nnz_per_interval.restype = ctypes.c_long
    L = _lib_expv
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    g0, gj = _colmajor(G0, Gj)
    o = lay.x_off if x_off is None else x_off
    kc = lay.K - k_first if k_count is None else k_count
    per = L.expv_ref_jac_nnz_per_interval(lay.d, lay.m)
    delta = np.full((lay.K, lay.x_dim), np.nan)
    jac = np.full((lay.K, per), np.nan) if want_jac else None
    off, terms = ctypes.c_double(0.0), ctypes.c_long(0)
    rc = L.expv_ref_eval_jac(lay.d, lay.m, lay.N, lay.z_dim, o, lay.u_off, lay.dt_off, _p(g0), _p(gj), _p(Z), _p(delta), _p(jac), nthreads, k_first, kc, chunk,
                             ctypes.byref(off), ctypes.byref(terms))
    assert rc == 0
    return delta, jac, {"off_structure_max": off.value, "taylor_terms": terms.value}

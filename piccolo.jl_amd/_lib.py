"""ctypes binding of csrc/libpiccolo_hip.so (C ABI: include/piccolo_hip.h).

The library is built in-tree by ``build_library`` (called from
``__graft_entry__.build()``) with ``hipcc --offload-arch=gfx950``.  There is no
CPU path: if the shared object is missing, or no gfx950 device is present when a
context is created, the error is raised to the caller -- nothing falls back.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libpiccolo_hip.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

PCL_OK = 0
PCL_EINVAL, PCL_ENOMEM, PCL_EHIP, PCL_ERCCL, PCL_ESHAPE, PCL_ENOTIMPL, PCL_EINTERNAL = -1, -2, -3, -4, -5, -6, -7
PCL_BATCH_MEMBERS, PCL_BATCH_TRAJ = 0, 1
PCL_STATE_VECTOR = -1  # pcl_desc.state_cols: general real d x d generator on one real column (compact density vectors)
_STATUS_NAMES = {0: "PCL_OK", -1: "PCL_EINVAL", -2: "PCL_ENOMEM", -3: "PCL_EHIP", -4: "PCL_ERCCL", -5: "PCL_ESHAPE", -6: "PCL_ENOTIMPL", -7: "PCL_EINTERNAL"}

# every symbol include/piccolo_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "pcl_create", "pcl_destroy", "pcl_last_error", "pcl_version",
    "pcl_constraint_dim", "pcl_jac_nnz", "pcl_hess_nnz",
    "pcl_jac_structure", "pcl_jac_structure_i64", "pcl_hess_structure", "pcl_hess_structure_i64",
    "pcl_eval", "pcl_jac", "pcl_eval_jac", "pcl_hess",
    "pcl_set_stream", "pcl_reset_stream", "pcl_sync", "pcl_eval_dev", "pcl_eval_jac_dev", "pcl_hess_dev",
    "pcl_jac_compact_nnz", "pcl_eval_jac_compact_dev", "pcl_jac_expand_dev",
    "pcl_deriv_nnz", "pcl_deriv_structure", "pcl_deriv_eval_jac", "pcl_deriv_eval_jac_dev",
    "pcl_jac_dev", "pcl_set_member_window", "pcl_set_goal", "pcl_set_goal_subspace", "pcl_set_weights", "pcl_infidelity_dev", "pcl_add_regularizer",
    "pcl_clear_regularizers", "pcl_objective_dev", "pcl_objective", "pcl_merit_grad_len", "pcl_merit_grad_dev", "pcl_eval_jac_merit_dev", "pcl_eval_jac_merit_objective_dev",
    "pcl_rollout", "pcl_rollout_dev",
    "pcl_comm_get_unique_id", "pcl_comm_init", "pcl_reduce_sum_dev", "pcl_reduce_sum", "pcl_comm_destroy",
    "pcl_set_option", "pcl_get_option", "pcl_codegen_source",
    "pcl_codegen_source_v4", "pcl_jit_prebuild", "pcl_set_order_policy", "pcl_set_order_from_trajectory", "pcl_order_for_bounds",
    "pcl_set_goal_form", "pcl_objective_hess_nnz", "pcl_objective_hess_structure", "pcl_objective_hess_dev", "pcl_objective_hess",
]  # fmt: skip


# include/piccolo_hip_lab.h: lab builds only (-DPCL_LAB -> libpiccolo_hip_lab.so); the shipped library must NOT export these
LAB_EXPORTS = ["pcl_resident_start", "pcl_resident_post", "pcl_resident_wait", "pcl_resident_stop", "pcl_resident_completed", "pcl_resident_stamps",
               "pcl_debug_timing", "pcl_codegen_apply_v4"]  # fmt: skip
SO_PATH_LAB = os.path.join(CSRC, "libpiccolo_hip_lab.so")


class PclError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (_STATUS_NAMES.get(code, "PCL_E?"), code, msg))
        self.code = code


class pcl_desc(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("d", ctypes.c_int32),
        ("n_drives", ctypes.c_int32),
        ("N", ctypes.c_int32),
        ("z_dim", ctypes.c_int32),
        ("u_off", ctypes.c_int32),
        ("dt_off", ctypes.c_int32),
        ("batch", ctypes.c_int32),
        ("batch_mode", ctypes.c_int32),
        ("pade_order", ctypes.c_int32),
        ("device_id", ctypes.c_int32),
        ("index_base", ctypes.c_int32),
        ("per_member_G0", ctypes.c_int32),
        ("state_cols", ctypes.c_int32),
        ("global_dim", ctypes.c_int64),
        ("G0", ctypes.POINTER(ctypes.c_double)),
        ("Gj", ctypes.POINTER(ctypes.c_double)),
        ("x_offs", ctypes.POINTER(ctypes.c_int32)),
    ]


def _source_digest(paths, flags):
    import hashlib

    h = hashlib.sha256(" ".join(flags).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()


def _file_digest(path):
    import hashlib

    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def build_library(force=False, verbose=False, profile=False, lab=False):
    """Compile csrc/piccolo_hip.hip for gfx950 into csrc/libpiccolo_hip.so (in-tree).

    The library is rebuilt whenever the digest of its sources and flags differs from the one recorded next to the
    shared object (``libpiccolo_hip.so.digest``: source digest + digest of the binary) -- modification times play no part, so a stale binary is never
    reused after an edit, a checkout or a copy to another box.  ``lab=True`` adds ``-DPCL_LAB`` (the entry points of include/piccolo_hip_lab.h),
    ``profile=True`` adds ``-DPCL_PROFILE -DPCL_LAB`` (cycle stamps inside the kernels for lab/probes); both write ``libpiccolo_hip_lab.so`` --
    never the shipped file -- and make this process's ``load()`` take that library."""
    global _variant_path
    lab = lab or profile
    out = SO_PATH_LAB if lab else SO_PATH
    src = os.path.join(CSRC, "piccolo_hip.hip")
    deps = [src, os.path.join(INCLUDE, "piccolo_hip.h"), os.path.join(INCLUDE, "piccolo_hip_lab.h")] + sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))  # the kernel families are headers of this one TU
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread"] + (["-DPCL_LAB"] if lab else []) + (["-DPCL_PROFILE"] if profile else [])
    digest = _source_digest(deps, flags)
    stamp = out + ".digest"
    if lab:
        if _lib is not None and _variant_path != out:
            raise RuntimeError("the shipped library is already loaded in this process: build the lab variant before the first load()")
        _variant_path = out
    if not force and os.path.exists(out) and os.path.exists(stamp):
        with open(stamp) as f:
            rec = f.read().split()
        # the record names the sources AND the binary built from them: a stamp restored by a checkout next to a binary built
        # from other sources (it happened: an experiment survived its revert) does not pass
        if len(rec) == 2 and rec[0] == digest and rec[1] == _file_digest(out):
            return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        raise FileNotFoundError("%s not found and %s is stale or missing (sources changed since it was built)" % (hipcc, out))
    cmd = [hipcc] + flags + ["-I", INCLUDE, "-o", out, src]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(digest + " " + _file_digest(out) + "\n")
    return out


_variant_path = None  # set by build_library(lab=True / profile=True): what load() opens in this process
_lib = None


def load():
    """dlopen the product library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = _variant_path or SO_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path
        )
    L = ctypes.CDLL(path)
    c_i64p = ctypes.POINTER(ctypes.c_int64)
    c_i32p = ctypes.POINTER(ctypes.c_int32)
    vp = ctypes.c_void_p
    L.pcl_create.argtypes = [ctypes.POINTER(pcl_desc), ctypes.POINTER(vp)]
    L.pcl_destroy.argtypes = [vp]
    L.pcl_destroy.restype = None
    L.pcl_last_error.argtypes = [vp]
    L.pcl_last_error.restype = ctypes.c_char_p
    L.pcl_version.restype = ctypes.c_char_p
    L.pcl_constraint_dim.argtypes = [vp, c_i64p, c_i64p, c_i64p]
    for f in ("pcl_jac_nnz", "pcl_hess_nnz", "pcl_jac_compact_nnz"):
        getattr(L, f).argtypes = [vp, c_i64p, c_i64p]
    for f in ("pcl_jac_structure", "pcl_hess_structure"):
        getattr(L, f).argtypes = [vp, c_i32p, c_i32p]
    for f in ("pcl_jac_structure_i64", "pcl_hess_structure_i64"):
        getattr(L, f).argtypes = [vp, c_i64p, c_i64p]
    # data pointers are passed as integers (host numpy .ctypes.data or device data_ptr())
    L.pcl_eval.argtypes = [vp, vp, vp]
    L.pcl_jac.argtypes = [vp, vp, vp]
    L.pcl_eval_jac.argtypes = [vp, vp, vp, vp]
    L.pcl_hess.argtypes = [vp, vp, vp, vp]
    L.pcl_set_stream.argtypes = [vp, vp]
    L.pcl_reset_stream.argtypes = [vp]
    L.pcl_sync.argtypes = [vp]
    L.pcl_eval_dev.argtypes = [vp, vp, vp]
    L.pcl_eval_jac_dev.argtypes = [vp, vp, vp, vp]
    L.pcl_hess_dev.argtypes = [vp, vp, vp, vp]
    L.pcl_eval_jac_compact_dev.argtypes = [vp, vp, vp, vp]
    L.pcl_jac_expand_dev.argtypes = [vp, vp, vp]
    L.pcl_deriv_nnz.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, c_i64p, c_i64p]
    L.pcl_deriv_structure.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_i64p, c_i64p]
    L.pcl_deriv_eval_jac.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp, vp]
    L.pcl_deriv_eval_jac_dev.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp, vp]
    L.pcl_set_goal.argtypes = [vp, vp]
    L.pcl_jac_dev.argtypes = [vp, vp, vp]
    L.pcl_set_member_window.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]
    L.pcl_set_goal_subspace.argtypes = [vp, vp, c_i32p, ctypes.c_int32]
    L.pcl_set_weights.argtypes = [vp, vp]
    L.pcl_add_regularizer.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int32]
    L.pcl_clear_regularizers.argtypes = [vp]
    L.pcl_objective_dev.argtypes = [vp, vp, ctypes.c_double, vp, vp]
    L.pcl_objective.argtypes = [vp, vp, ctypes.c_double, vp, vp]
    L.pcl_merit_grad_len.argtypes = [vp, c_i64p, c_i64p]
    L.pcl_merit_grad_dev.argtypes = [vp, vp, vp, vp, vp]
    L.pcl_eval_jac_merit_dev.argtypes = [vp, vp, vp, vp, vp, vp]
    L.pcl_eval_jac_merit_objective_dev.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_double, vp, vp]
    L.pcl_infidelity_dev.argtypes = [vp, vp, ctypes.c_double, vp, vp]
    L.pcl_rollout.argtypes = [vp, vp, vp]
    L.pcl_rollout_dev.argtypes = [vp, vp, vp]
    L.pcl_comm_get_unique_id.argtypes = [ctypes.c_char_p]
    L.pcl_comm_init.argtypes = [vp, ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32]
    L.pcl_reduce_sum_dev.argtypes = [vp, vp, ctypes.c_int64]
    L.pcl_reduce_sum.argtypes = [vp, vp, ctypes.c_int64]
    L.pcl_comm_destroy.argtypes = [vp]
    L.pcl_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
    L.pcl_get_option.argtypes = [vp, ctypes.c_char_p, c_i64p]
    L.pcl_codegen_source.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_char_p, ctypes.c_int64, c_i64p]
    L.pcl_codegen_source_v4.argtypes = [ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, c_i64p]
    if path == SO_PATH_LAB:  # include/piccolo_hip_lab.h
        L.pcl_resident_start.argtypes = [vp, vp, vp, vp]
        L.pcl_resident_post.argtypes = [vp, ctypes.c_int32]
        L.pcl_resident_wait.argtypes = [vp, ctypes.c_double]
        L.pcl_resident_stop.argtypes = [vp]
        L.pcl_resident_completed.argtypes = [vp, c_i64p]
        L.pcl_resident_stamps.argtypes = [vp, vp, ctypes.c_int64]
        L.pcl_debug_timing.argtypes = [vp, c_i64p, ctypes.c_int64]
        L.pcl_codegen_apply_v4.argtypes = [ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int]
    _lib = L
    return L


def prebuild_kernels(G_drifts, G_drives, orders=(4,), hessian=True, out_dir=None, resident=False):  # (resident: lab builds only)
    """Compile the pattern-compiled modules a context of this system (one drift, or the per-member drifts of an ensemble) would compile
    with hiprtc on first use, into ``csrc/prebuilt/`` (or ``out_dir``) under their content hashes -- a fresh process (every rank of a job)
    then loads them instead of compiling.  No device needed.  Returns the number of modules asked for."""
    import numpy as np

    L = load()
    G0 = np.ascontiguousarray(np.stack([np.asarray(g, dtype=np.float64).T for g in (G_drifts if np.ndim(G_drifts) == 3 else [G_drifts])]))
    Gj = np.ascontiguousarray(np.stack([np.asarray(g, dtype=np.float64).T for g in G_drives])) if len(G_drives) else np.zeros((0, 1, 1))
    n = G0.shape[-1]
    d, m = n // 2, len(G_drives)
    od = out_dir.encode() if out_dir else None
    L.pcl_jit_prebuild.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    count = 0
    for order in orders:
        # 0 fused | 4 fused with the slice-ticket roles | 3 the order-4 Hessian module | 5 the column-group Hessian kernel (`auto` at the other orders)
        whats = [0] + ([4] if order <= 4 else []) + (([3, 5] if order == 4 else [5]) if hessian else [])  # (order 4: kernel 6, and the column-group kernel for one trajectory)
        whats += [6] if resident else []  # the resident evaluator's module
        for what in whats:
            rc = L.pcl_jit_prebuild(d, m, G0.ctypes.data, G0.shape[0], Gj.ctypes.data if m else None, order // 2, what, od)
            if rc != 0:
                raise PclError(rc, L.pcl_last_error(None).decode())
            count += 1
    return count

"""CPU tests of the drop-in boundary and the host-side logic (no GPU compute):
the C-ABI library loads and exports every symbol include/piccolo_hip.h declares,
argument validation is loud, no CPU fallback exists, host builders reproduce the
reference's known answers and layout."""
import ctypes
import os
import re

import numpy as np
import pytest

import piccolo_jl_amd as pa
from oracle import pade_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    pa.build_library()
    return pa._lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "piccolo_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char \*)\s*\*?\s*(pcl_\w+)\s*\(", hdr, flags=re.M))
    declared.discard("pcl_comm_id")
    assert declared, "no declarations parsed"
    assert declared == set(pa._lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pcl_version()
    assert len(declared) <= 62  # (round-5 review: 70 with the variants that lost and the debugging aids)
    # what lost or only debugs lives in include/piccolo_hip_lab.h and in lab builds (-DPCL_LAB): declared there, NOT exported by the shipped library
    lab_hdr = open(os.path.join(ROOT, "include", "piccolo_hip_lab.h")).read()
    lab_declared = set(re.findall(r"^(?:int|void|const char \*)\s*\*?\s*(pcl_\w+)\s*\(", lab_hdr, flags=re.M))
    assert lab_declared == set(pa._lib.LAB_EXPORTS) and not (lab_declared & declared)
    shipped = ctypes.CDLL(pa._lib.SO_PATH)
    for name in lab_declared:
        assert not hasattr(shipped, name), "%s is exported by the shipped library" % name


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "piccolo_hip.h"\nint main(void){ pcl_desc d; d.struct_size = (int)sizeof d; return d.struct_size == 0; }\n')
    import subprocess

    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_desc_struct_matches_header(lib):
    # sizeof(pcl_desc) in ctypes == what the library expects (it rejects a mismatch)
    d = pa._lib.pcl_desc()
    d.struct_size = ctypes.sizeof(pa._lib.pcl_desc) + 8
    h = ctypes.c_void_p()
    rc = lib.pcl_create(ctypes.byref(d), ctypes.byref(h))
    assert rc == pa._lib.PCL_EINVAL and b"ABI mismatch" in lib.pcl_last_error(None)
    assert ctypes.sizeof(pa._lib.pcl_desc) == 88


def _mk(**over):
    kw = dict(d=2, m=2, N=5, z_dim=16, u_off=10, dt_off=8, x_offs=[0], G0=np.zeros((4, 4)), Gj=np.zeros((2, 4, 4)), batch=1,
              batch_mode=pa._lib.PCL_BATCH_MEMBERS)  # fmt: skip
    kw.update(over)
    return kw


@pytest.mark.parametrize(
    "over,code,msg",
    [
        (dict(pade_order=7), pa._lib.PCL_ENOTIMPL, "orders 2, 4, 6, 8, 10"),
        (dict(d=33, G0=np.zeros((66, 66)), Gj=np.zeros((2, 66, 66)), z_dim=3000, x_offs=[0], u_off=2900, dt_off=2899), pa._lib.PCL_ESHAPE, "generator dimension"),
        (dict(N=1), pa._lib.PCL_EINVAL, "N>=2"),
        (dict(x_offs=[12]), pa._lib.PCL_EINVAL, "does not fit"),
        (dict(u_off=15), pa._lib.PCL_EINVAL, "u_off"),
        (dict(index_base=2), pa._lib.PCL_EINVAL, "index_base"),
        (dict(batch_mode=7), pa._lib.PCL_EINVAL, "batch_mode"),
    ],
)
def test_create_validates_before_touching_the_device(lib, over, code, msg):
    with pytest.raises(pa.PclError) as ei:
        pa.integrators._PclContext(**_mk(**over))
    assert ei.value.code == code and msg in str(ei.value)


def test_no_cpu_fallback(lib):
    """Without a GPU the product path fails loudly (PCL_EHIP); it never computes on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pa.PclError) as ei:
        pa.integrators._PclContext(**_mk())
    assert ei.value.code == pa._lib.PCL_EHIP and "no CPU path" in str(ei.value)
    s = pa.QuantumSystem(0.5 * pa.PAULIS["Z"], [pa.PAULIS["X"], pa.PAULIS["Y"]], [1.0, 1.0])
    t = pa.unitary_trajectory(s, np.zeros((2, 6)), np.linspace(0, 1, 6), pa.GATES["X"])
    with pytest.raises(pa.PclError):
        pa.BilinearIntegrator(s, t)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "piccolo.jl_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".jl", ".cpp")):
                txt = open(os.path.join(dp, fn), encoding="utf-8").read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dp, fn)


# ---- host logic: builders and layout ------------------------------------------------------------
def test_product_builders_match_reference_literals():
    assert np.allclose(pa.quantum.G(np.array([[1, 2], [3, 4]]) + 1j * np.array([[0, 1], [1, 0]])), [[0, 1, 1, 2], [1, 0, 3, 4], [-1, -2, 0, 1], [-3, -4, 1, 0]])
    assert np.allclose(pa.operator_to_iso_vec(np.array([[0, 1 - 1j], [1 + 1j, 0]])), [0, 1, 0, 1, 1, 0, -1, 0])
    assert np.allclose(pa.iso_vec_to_operator([0, 1, 0, 1, 1, 0, -1, 0]), [[0, 1 - 1j], [1 + 1j, 0]])
    assert np.allclose(pa.annihilate(3), [[0, 1, 0], [0, 0, np.sqrt(2)], [0, 0, 0]])
    assert np.allclose(pa.quantum.iso_vec_to_iso_operator([0, 1, 0, 1, 1, 0, -1, 0]), [[0, 1, 0, 1], [1, 0, -1, 0], [0, -1, 0, 1], [1, 0, 1, 0]])


@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_product_systems_match_oracle_systems(cfg):
    o = po.config_system(cfg)
    if cfg == 1:
        s = pa.QuantumSystem(0.5 * pa.PAULIS["Z"], [pa.PAULIS["X"], pa.PAULIS["Y"]], [1.0, 1.0])
    elif cfg == 2:
        s = pa.MultiTransmonSystem([4.0, 4.1], [0.2, 0.2], [[0, 0.1], [0.1, 0]], levels_per_transmon=2, drive_bounds=0.1)
    else:
        s = pa.MultiTransmonSystem([4.0, 4.1, 4.2], [0.2, 0.21, 0.22], [[0, 0.01, 0.02], [0.01, 0, 0.03], [0.02, 0.03, 0]], drive_bounds=0.1)
    assert s.levels == o.levels and s.n_drives == o.n_drives
    assert np.allclose(s.G_drift, o.G_drift, rtol=0, atol=1e-13)
    assert np.allclose(s.G_drives_array(), np.array(o.G_drives), rtol=0, atol=1e-13)
    assert s.drive_bounds == o.drive_bounds
    # Hermitian H  <=>  skew-symmetric G
    assert np.allclose(s.G_drift, -s.G_drift.T)


def test_named_trajectory_layout():
    s = pa.MultiTransmonSystem([4.0, 4.1], [0.2, 0.2], [[0, 0.1], [0.1, 0]], levels_per_transmon=2, drive_bounds=0.1)
    N = 7
    t = pa.unitary_trajectory(s, 0.01 * np.arange(4 * N).reshape(4, N), np.linspace(0, 1, N), pa.GATES["CX"])
    assert t.names == ("Ũ⃗", "Δt", "t", "u", "du", "ddu") and t.dim == 46 and t.N == N
    assert t.components["Ũ⃗"] == range(0, 32) and t.components["Δt"] == range(32, 33) and t.components["u"] == range(34, 38)
    # knot-major: knot k is contiguous; data is a dim x N view
    assert np.array_equal(t.knot(3), t.data[:, 3]) and t.datavec.size == 46 * N
    assert np.array_equal(t["Ũ⃗"][:, 0], pa.operator_to_iso_vec(np.eye(4)))
    assert np.allclose(t["du"][:, 0], (t["u"][:, 1] - t["u"][:, 0]) / t["Δt"][0, 0])
    st = pa.sampling_trajectory([s, s, s], np.zeros((4, N)), np.linspace(0, 1, N), pa.GATES["CX"])
    assert st.names == ("Ũ⃗1", "Ũ⃗2", "Ũ⃗3", "Δt", "t", "u") and st.dim == 3 * 32 + 2 + 4
    with pytest.raises(ValueError):
        pa.NamedTrajectory({"a": np.zeros((2, 3)), "b": np.zeros((1, 4))})


def test_quantum_system_validation():
    with pytest.raises(AssertionError):
        pa.QuantumSystem(np.array([[0, 1], [0, 0]]), [pa.PAULIS["X"]], [1.0])
    with pytest.raises(AssertionError):
        pa.QuantumSystem(pa.PAULIS["Z"], [np.array([[0, 1], [0, 0]])], [1.0])
    with pytest.raises(ValueError):
        pa.lift_operator(pa.PAULIS["X"], 1, [3, 2])


def test_objective_terms_and_subspace_helpers_host_side():
    """Host-side records of the objective mirror (no GPU): term composition, EmbeddedOperator, subspace indices
    [REF src/quantum/operators/embedded_operators.jl:627-633 literals, 1-based there]."""
    assert [i + 1 for i in pa.get_subspace_indices([[0, 1], [0, 1]], [3, 3])] == [1, 2, 4, 5]
    assert [i + 1 for i in pa.get_subspace_indices([[1], [0, 1]], [3, 3])] == [4, 5]
    X = pa.PAULIS["X"]
    op = pa.EmbeddedOperator(X, [0, 1], 3)
    assert op.operator.shape == (3, 3) and np.array_equal(op.unembed(), X.astype(complex)) and op.operator[2, 2] == 0
    with pytest.raises(ValueError):
        pa.EmbeddedOperator(X, [0, 1, 2], 3)
    s = pa.QuantumSystem(0.5 * pa.PAULIS["Z"], [pa.PAULIS["X"], pa.PAULIS["Y"]], [1.0, 1.0])
    t = pa.unitary_trajectory(s, np.zeros((2, 6)), np.linspace(0, 1, 6), pa.GATES["X"])
    J = pa.UnitaryInfidelityObjective(pa.GATES["X"], "Ũ⃗", t, Q=50.0) + pa.QuadraticRegularizer("u", t, 1e-2) + pa.QuadraticRegularizer("du", t, [1.0, 2.0], 0)
    assert isinstance(J, pa.Objective) and len(J.terms) == 3
    assert J.terms[1].off == t.components["u"].start and J.terms[2].dt_power == 0 and np.array_equal(J.terms[2].R, [1.0, 2.0])
    with pytest.raises(RuntimeError):
        J.value_and_gradient(t)  # not bound to a context: nothing is ever computed on the host


def test_config4_synthetic_members_match_the_oracle_definition():
    """piccolo.jl_amd/synthetic.py (product side, used by bench.py) against SURVEY 8(d)'s definition restated with the oracle:
    H_drift_i = H_drift + eps_i 2 pi sum_q a_q' a_q, eps_i ~ U(-1e-3, 1e-3), default_rng(2000 + i)."""
    from oracle import pade_oracle as po
    from piccolo_jl_amd import synthetic

    base = po.config_system(3)
    a = po.annihilate(3)
    num = sum(po.lift_operator(a.conj().T @ a, q, [3, 3, 3]) for q in (1, 2, 3))
    members = synthetic.config4_members(5, 3)
    for j, s in enumerate(members):
        eps = np.random.default_rng(2000 + 5 + j).uniform(-1e-3, 1e-3)
        ref = po.System(base.H_drift + eps * 2 * np.pi * num, base.H_drives, base.drive_bounds)
        assert np.allclose(s.G_drift, ref.G_drift, rtol=0, atol=1e-13)
        assert np.allclose(s.G_drives_array(), np.array(ref.G_drives), rtol=0, atol=1e-13)
    tr = synthetic.synthetic_ensemble(members[:2], 4, seed=1)
    assert list(tr.components)[:3] == ["Ũ⃗1", "Ũ⃗2", "Δt"] and tr.dim == 2 * 1458 + 2 + 18


def _generated_source(lib, G0, Gj):
    import ctypes
    n = G0.shape[0]
    g0 = np.ascontiguousarray(G0.T).ravel()  # column-major
    gj = np.ascontiguousarray(np.stack([g.T for g in Gj])).ravel()
    need = ctypes.c_int64()
    assert lib.pcl_codegen_source(n // 2, len(Gj), g0.ctypes.data, gj.ctypes.data, None, 0, ctypes.byref(need)) == 0
    buf = ctypes.create_string_buffer(need.value)
    assert lib.pcl_codegen_source(n // 2, len(Gj), g0.ctypes.data, gj.ctypes.data, buf, need.value, ctypes.byref(need)) == 0
    return buf.value.decode()


def test_pattern_compiled_kernel_source(lib, tmp_path):
    """The source the library generates for BASELINE config 3 (and compiles with hiprtc on first use): deterministic, one
    multiply-add statement per entry of the union pattern of G(u) in the big product, and it compiles for gfx950 with the
    toolchain of this image (no GPU needed) into a kernel that fits the register file without scratch traffic worth noting."""
    import re, shutil, subprocess
    from piccolo_jl_amd import synthetic
    s3 = synthetic.config_system(3)
    Gj = s3.G_drives_array()
    src = _generated_source(lib, s3.G_drift, Gj)
    assert src == _generated_source(lib, s3.G_drift, Gj)
    assert "pcl_kernel_hessian_sparse.hpp" in src and "#define SPD 27" in src and "#define SPM 6" in src
    n, d = 54, 27
    union = s3.G_drift[:, :d] != 0
    for g in Gj:
        union |= g[:, :d] != 0
    nz = int(union.sum())
    assert "#define SPNZ %d\n" % nz in src
    for fn, end in (("void sp_gt(", "void sp_g("), ("void sp_g(", "struct sp_mags")):  # G(u)^T x and G(u) x
        body = src[src.index(fn):src.index(end)]
        assert len(re.findall(r"v_(?:mul|fmac|fma)_f64", body)) == nz  # one instruction per entry: no padding, no dense tiles
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    f = tmp_path / "sp.hip"
    f.write_text(src)
    csrc = os.path.join(os.path.dirname(pa.__file__), "csrc") if os.path.isdir(os.path.join(os.path.dirname(pa.__file__), "csrc")) else None
    assert csrc is not None
    out = tmp_path / "sp.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-include", "hip/hip_runtime.h", "-I", csrc, "-S",
                        "--cuda-device-only", "-o", str(out), str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert "pcl_hess_sparse_kernel" in asm and "pcl_sparse_values_kernel" in asm and "pcl_eval_sparse_kernel" in asm
    scratch = [int(x) for x in re.findall(r"; ScratchSize: (\d+)", asm)]
    assert max(scratch) <= 256  # a few loop-invariant integers, not operand arrays


def test_pattern_compiled_fused_and_hessian_sources(lib, tmp_path):
    """Kernel 4 and its relatives (fused residual + Jacobian, residual only, general-order Hessian): the generator's term tables
    reproduce G(u) x and G(u)^T x on the host, the source is deterministic, holds exactly one multiply-add per entry of the left
    column block of the union pattern per product, reads no per-interval table at BASELINE config 3 (every drift value class is
    resident) and streams the less used classes for an ensemble's members; it compiles for gfx950 here (no GPU needed) with
    the fused kernel inside its 128 registers per lane without scratch traffic worth noting."""
    import ctypes, re, shutil, subprocess
    from piccolo_jl_amd import synthetic
    s3 = synthetic.config_system(3)
    Gj = s3.G_drives_array()
    n = s3.G_drift.shape[0]
    d, m = n // 2, len(Gj)
    gj = np.ascontiguousarray(np.stack([g.T for g in Gj])).ravel()

    def source(G0s, q, what):
        g0 = np.ascontiguousarray(np.stack([g.T for g in G0s])).ravel()
        need = ctypes.c_int64()
        assert lib.pcl_codegen_source_v4(d, m, g0.ctypes.data, len(G0s), gj.ctypes.data, q, what, None, 0, ctypes.byref(need)) == 0
        buf = ctypes.create_string_buffer(need.value)
        assert lib.pcl_codegen_source_v4(d, m, g0.ctypes.data, len(G0s), gj.ctypes.data, q, what, buf, need.value, ctypes.byref(need)) == 0
        return buf.value.decode()

    g0 = np.ascontiguousarray(s3.G_drift.T).ravel()
    rng = np.random.default_rng(0)
    # the term tables applied on the host: pcl_codegen_v4.hpp is host-only code, compiled here with g++ behind the lab header's signature
    # (the entry point itself is in lab builds of the library only: include/piccolo_hip_lab.h)
    shim = tmp_path / "apply_shim.cpp"
    shim.write_text('#include "pcl_codegen_v4.hpp"\n'
                    'extern "C" int pcl_codegen_apply_v4(int d, int m, const double *G0, int n_g0, const double *Gj, const double *u, const double *x, double *y, int transposed) {\n'
                    '    const pcl_codegen::V4Plan plan = pcl_codegen::make_v4_plan(d, m, G0, n_g0, Gj);\n'
                    '    if (!plan.ok) return -5;\n'
                    '    if (transposed) pcl_codegen::v4_reference_apply_t(plan, G0, u, x, y); else pcl_codegen::v4_reference_apply(plan, G0, Gj, u, x, y);\n'
                    '    return 0;\n}\n')
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "piccolo.jl_amd", "csrc"), "-o", str(tmp_path / "apply_shim.so"), str(shim)])
    shim_lib = ctypes.CDLL(str(tmp_path / "apply_shim.so"))
    shim_lib.pcl_codegen_apply_v4.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int]
    for tr in (0, 1):
        for _ in range(3):
            u, x, y = rng.normal(size=m), rng.normal(size=n), np.zeros(n)
            assert shim_lib.pcl_codegen_apply_v4(d, m, g0.ctypes.data, 1, gj.ctypes.data, u.ctypes.data, x.ctypes.data, y.ctypes.data, tr) == 0
            G = s3.G_drift + np.tensordot(u, Gj, axes=1)
            assert np.abs((G.T if tr else G) @ x - y).max() < 1e-13
    union = s3.G_drift[:, :d] != 0
    for g in Gj:
        union |= g[:, :d] != 0
    nz = int(union.sum())
    src = source([s3.G_drift], 4, 0)
    assert src == source([s3.G_drift], 4, 0)
    assert "#define SP4Q 4" in src and "pcl_kernel_fused_sparse.hpp" in src
    pat = r"v_(?:mul|fmac|fma)_f64 %\[a[UV][^\n]*%\[x[0-9]+\]"
    for fn, end in (("void sp4_product(", "void sp4_product0("), ("void sp4_product0(", "#define SP4_COOP")):
        body = src[src.index(fn):src.index(end)]
        assert len(re.findall(pat, body)) == nz  # one instruction per entry: no padding, no dense tiles
        assert "s_load" not in body  # every coefficient resident: nothing is read inside the product
    # the cooperative first item: four row ranges of the product without Y, every row with the instruction sequence it has in the whole
    assert "#define SP4_COOP 1" in src and "#define SP4_NPART 4" in src
    whole = re.findall(pat, src[src.index("void sp4_product0("):src.index("#define SP4_COOP")])
    parts = []
    for k in range(4):
        a, b = src.index("void sp4_product0_p%d(" % k), src.index("void sp4_product_p%d(" % k)  # without Y | with Y (the cooperative residual kernel)
        e = src.index("void sp4_product0_p%d(" % (k + 1)) if k < 3 else src.index("void sp4_product0_part(")
        parts.append(re.findall(pat, src[a:b]))
        assert len(parts[-1]) > nz // 8 and re.findall(pat, src[b:e]) == parts[-1]
    strip = lambda lines: [re.sub(r"a([UV])[01]_", r"a\1_", x) for x in lines]  # (the accumulator set alternates with the group's position)
    assert strip(sum(parts, [])) == strip(whole)
    members = synthetic.config4_members(0, 3)
    srcE = source([s.G_drift for s in members], 2, 0)
    assert "s_load_dwordx16" in srcE  # 27 value classes: the less used ones are streamed
    srcH = source([s3.G_drift], 4, 1)
    assert "pcl_kernel_hess_sparse4.hpp" in srcH and "void sp4_product_t(" in srcH and "void sp4_gdot_all(" in srcH and "sp4_gtv" not in srcH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(os.path.dirname(pa.__file__), "csrc")
    for name, text, kernels in (("fused", src, ("pcl_fused_sparse_kernel", "pcl_eval_sparse4_kernel")), ("fusedE", srcE, ("pcl_fused_sparse_kernel",)),
                                ("hess", srcH, ("pcl_hess_sparse4_kernel",))):
        f = tmp_path / (name + ".hip")
        f.write_text(text)
        out = tmp_path / (name + ".s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-include", "hip/hip_runtime.h", "-I", csrc, "-S", "--cuda-device-only",
                            "-o", str(out), str(f)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = out.read_text()
        for k in kernels:
            assert k in asm
        if name != "hess":
            scratch = [int(x) for x in re.findall(r"; ScratchSize: (\d+)", asm)]
            assert max(scratch) <= 64


def test_order_policy_without_a_device(lib):
    """pcl_order_for_bounds (what pcl_set_order_policy applies): theta is the exact maximum of |G(u)|_2 over the box of controls -- a vertex --
    times dt_max, never the triangle bound; a generator whose rows sum to zero is not mistaken for zero (round-4 ADVICE: power iteration from the
    all-ones vector); BASELINE config 3's bounds give order 10 at 1e-10 and order 8 at the synthetic trajectories' own scale; a tolerance no order
    up to 10 meets is reported."""
    import ctypes
    import itertools
    import math

    from piccolo_jl_amd import synthetic

    lib.pcl_order_for_bounds.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                         ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    kappa = lambda q: math.factorial(q) ** 2 / (math.factorial(2 * q) * math.factorial(2 * q + 1))
    want = lambda th, tol: next((2 * q for q in range(1, 6) if kappa(q) * th ** (2 * q + 1) <= tol), 10)

    def policy(G0s, Gj, dt, um, tol):
        G0s = np.atleast_3d(np.asarray(G0s, dtype=np.float64).reshape(-1, G0s.shape[-1], G0s.shape[-1]))
        n, m = G0s.shape[-1], len(Gj)
        g0 = np.ascontiguousarray(np.stack([g.T for g in G0s]))
        gj = np.ascontiguousarray(np.stack([g.T for g in Gj])) if m else np.zeros(1)
        u = np.ascontiguousarray(np.broadcast_to(np.asarray(um, dtype=np.float64), (max(m, 1),)))
        th, od, met = ctypes.c_double(), ctypes.c_int32(), ctypes.c_int32()
        rc = lib.pcl_order_for_bounds(n, m, g0.ctypes.data, len(G0s), gj.ctypes.data, dt, u.ctypes.data, tol, ctypes.byref(th), ctypes.byref(od), ctypes.byref(met))
        assert rc == 0
        return th.value, od.value, met.value

    s3 = synthetic.config_system(3)
    G0, Gj = s3.G_drift, s3.G_drives_array()
    box = lambda um: 0.1 * max(np.linalg.norm(G0 + np.tensordot(um * np.array(sg), Gj, axes=1), 2) for sg in itertools.product((-1.0, 1.0), repeat=len(Gj)))
    tri = 0.1 * (np.linalg.norm(G0, 2) + 0.1 * sum(np.linalg.norm(g, 2) for g in Gj))
    th, od, met = policy(G0, Gj, 0.1, 0.1, 1e-10)
    assert abs(th - box(0.1)) <= 1e-8 * th and th < 0.7 * tri and (od, met) == (10, 1) == (want(th, 1e-10), 1)
    th, od, met = policy(G0, Gj, 0.1, 0.02, 1e-10)
    assert abs(th - box(0.02)) <= 1e-8 * th and (od, met) == (8, 1)
    assert policy(G0, Gj, 0.1, 0.1, 1e-13)[1:] == (10, 0)  # kappa_5 theta^11 = 1.6e-12 > 1e-13: said, not hidden
    assert policy(G0, Gj, 0.01, 0.1, 1e-10)[1] == want(0.1 * box(0.1), 1e-10)
    # per-member drifts: theta covers every member (the bound used: vertex maximum of member 0 + |G0_b - G0_0|)
    members = synthetic.config4_members(0, 3)
    th_m = policy(np.stack([s.G_drift for s in members]), Gj, 0.1, 0.1, 1e-10)[0]
    exact = max(0.1 * max(np.linalg.norm(s.G_drift + np.tensordot(0.1 * np.array(sg), Gj, axes=1), 2) for sg in itertools.product((-1.0, 1.0), repeat=len(Gj))) for s in members)
    assert exact <= th_m * (1 + 1e-12) and th_m <= exact * 1.01
    # rows that sum to zero
    H = 3.0 * np.array([[1.0, -1.0], [-1.0, 1.0]], dtype=complex)
    sz = pa.QuantumSystem(H, [pa.PAULIS["X"]], [1.0])
    th, od, _ = policy(sz.G_drift, sz.G_drives_array(), 0.1, 0.0, 1e-10)
    true = 0.1 * np.linalg.norm(sz.G_drift, 2)
    assert abs(th - true) <= 1e-8 * true and od == want(true, 1e-10)
    # no drives
    th, od, _ = policy(sz.G_drift, [], 0.2, 0.0, 1e-6)
    assert abs(th - 2 * true) <= 1e-8 * true and od == want(2 * true, 1e-6)


def test_bounds_formats_of_the_order_policy():
    """The order policy reads max(|lower|, |upper|) per drive from the trajectory's bounds in every shape the mirror and the reference hand them over in."""
    from piccolo_jl_amd.integrators import _abs_bound

    assert np.array_equal(_abs_bound((-0.1 * np.ones(6), 0.2 * np.ones(6)), 6), np.full(6, 0.2))  # (lower_vec, upper_vec): NamedTrajectory.bounds
    assert np.array_equal(_abs_bound([(-0.1, 0.2)] * 6, 6), np.full(6, 0.2))  # system.drive_bounds: a pair per drive
    assert np.array_equal(_abs_bound([(-0.3, 0.1), (-0.1, 0.2)], 2), [0.3, 0.2])
    assert np.array_equal(_abs_bound((np.array([-0.3, -0.1]), np.array([0.2, 0.4])), 2), [0.3, 0.4])
    assert np.array_equal(_abs_bound(0.3, 3), np.full(3, 0.3)) and np.array_equal(_abs_bound((0.05, 0.1), 1), [0.1])
    assert np.array_equal(_abs_bound(np.array([0.1, 0.2, 0.3]), 3), [0.1, 0.2, 0.3])
    with pytest.raises(ValueError):
        _abs_bound(np.zeros((3, 3)), 4)

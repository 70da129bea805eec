#!/usr/bin/env python3
"""Writes gen_cfg3_q4.inc (the generated functions of BASELINE config 3 at order 8, Hessian set, without their includes) and gen_noadd.inc (the same
without the products' ds_add_f64: WRONG results, timing only) next to this file; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -o prodbench prodbench.hip      (and prodbench_noadd from a copy that includes gen_noadd.inc)"""
import ctypes, os, sys
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(here))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import _lib, synthetic
pa.build_library()
L = _lib.load()
s = synthetic.config_system(3)
G0 = np.ascontiguousarray(np.asarray(s.G_drift).T[None])
Gj = np.ascontiguousarray(np.stack([np.asarray(g).T for g in s.G_drives_array()]))
need = ctypes.c_int64()
L.pcl_codegen_source_v4.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p]
L.pcl_codegen_source_v4(27, 6, G0.ctypes.data, 1, Gj.ctypes.data, 4, 1, None, 0, ctypes.byref(need))
buf = ctypes.create_string_buffer(need.value)
L.pcl_codegen_source_v4(27, 6, G0.ctypes.data, 1, Gj.ctypes.data, 4, 1, buf, need.value, ctypes.byref(need))
lines = [l for l in buf.value.decode().splitlines() if not l.startswith("#include")]
open(os.path.join(here, "gen_cfg3_q4.inc"), "w").write("\n".join(lines) + "\n")
open(os.path.join(here, "gen_noadd.inc"), "w").write("\n".join(l for l in lines if "ds_add_f64" not in l) + "\n")
print(len(lines), "lines")

#!/usr/bin/env python3
"""Registers, spills, LDS and text size of the pattern-compiled modules of a BASELINE config as the library compiles them (no GPU needed):
module_stats.py [config=3] [order=4] [what=0 ...]   (what: pcl_jit_prebuild's; 0 fused lean, 4 fused with ticket roles, 1/2/3 Hessian)"""
import os, subprocess, sys, tempfile, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import numpy as np
import piccolo_jl_amd as pa
from piccolo_jl_amd import _lib, synthetic

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
order = int(sys.argv[2]) if len(sys.argv) > 2 else 4
whats = [int(a) for a in sys.argv[3:]] or [0]
pa.build_library()
L = _lib.load()
s = synthetic.config_system(cfg)
G0 = np.ascontiguousarray(np.asarray(s.G_drift).T[None])
Gj = np.ascontiguousarray(np.stack([np.asarray(g).T for g in s.G_drives_array()]))
L.pcl_jit_prebuild.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
for what in whats:
    with tempfile.TemporaryDirectory() as td:
        rc = L.pcl_jit_prebuild(G0.shape[-1] // 2, Gj.shape[0], G0.ctypes.data, 1, Gj.ctypes.data, order // 2, what, td.encode())
        assert rc == 0, L.pcl_last_error(None).decode()
        for f in glob.glob(td + "/*.hsaco"):
            blob = open(f, "rb").read()
            n = int.from_bytes(blob[4:8], "little")  # "PCL2" | u32 name length | u64 code length | u64 checksum | name | code
            co = f + ".co"
            open(co, "wb").write(blob[24 + n:])
            notes = subprocess.run([readelf, "--notes", co], capture_output=True, text=True).stdout
            keep = [l.strip() for l in notes.splitlines() if any(k in l for k in (".vgpr_count", ".sgpr_count", "spill_count", ".group_segment_fixed_size", ".private_segment_fixed_size", ".name:"))]
            secs = subprocess.run([readelf, "-S", co], capture_output=True, text=True).stdout
            text = [l for l in secs.splitlines() if " .text " in l]
            if os.environ.get("KEEP"):
                open(os.environ["KEEP"], "wb").write(blob[24 + n:])  # the code object, for llvm-objdump -d
            print("what %d:" % what, "; ".join(keep))
            print("   ", text[0].split()[-6:] if text else "")

#!/usr/bin/env python3
"""Write the generated source of the pattern-compiled kernels of a BASELINE config (default 3) to a file (no GPU needed):
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I piccolo.jl_amd/csrc -S -o out.s <file> shows the ISA the library will run."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import piccolo_jl_amd as pa
from piccolo_jl_amd import _lib, synthetic

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/pcl_sparse_cfg%d.hip" % cfg
pa.build_library()
L = _lib.load()
s = synthetic.config_system(cfg)
n = s.G_drift.shape[0]
g0 = np.ascontiguousarray(s.G_drift.T).ravel()  # column-major
gj = np.ascontiguousarray(np.stack([g.T for g in s.G_drives_array()])).ravel()
need = ctypes.c_int64()
L.pcl_codegen_source(n // 2, len(s.G_drives_array()), g0.ctypes.data, gj.ctypes.data, None, 0, ctypes.byref(need))
buf = ctypes.create_string_buffer(need.value)
rc = L.pcl_codegen_source(n // 2, len(s.G_drives_array()), g0.ctypes.data, gj.ctypes.data, buf, need.value, ctypes.byref(need))
assert rc == 0
open(out, "w").write(buf.value.decode())
print(out, need.value, "bytes")

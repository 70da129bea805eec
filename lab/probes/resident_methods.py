"""The resident evaluator's Python methods (round 5: built, measured, loses -- DESIGN.md 4.2.2).  Its entry points exist in LAB builds of the
library only (include/piccolo_hip_lab.h, -DPCL_LAB), so the methods live here and not on the shipped context class:

    from piccolo_jl_amd import _lib
    _lib.build_library(lab=True)          # csrc/libpiccolo_hip_lab.so; this process's load() takes it
    import resident_methods               # (lab/probes on sys.path)
    resident_methods.attach()             # piccolo.jl_amd.integrators._PclContext gains resident_start / _post / _wait / _eval / _stop / _completed
"""
import ctypes

from piccolo_jl_amd import integrators as _integ
from piccolo_jl_amd.integrators import _ptr


# -- resident evaluator (include/piccolo_hip_lab.h: pcl_resident_*): eval_jac_dev as requests to workgroups that stay on the device
def resident_start(self, Z, delta, vals):
    self._chk(self._L.pcl_resident_start(self._h, _ptr(Z), _ptr(delta) if delta is not None else None, _ptr(vals)))
    self._resident_keep = (Z, delta, vals)  # the kernel holds their addresses

def resident_post(self, count=1):
    self._chk(self._L.pcl_resident_post(self._h, int(count)))

def resident_wait(self, timeout_s=10.0):
    self._chk(self._L.pcl_resident_wait(self._h, float(timeout_s)))

def resident_eval(self, timeout_s=10.0):
    """One evaluation of the trajectory array as it stands now: post + wait."""
    self.resident_post(1)
    self.resident_wait(timeout_s)

def resident_stop(self):
    self._chk(self._L.pcl_resident_stop(self._h))
    self._resident_keep = None

def resident_completed(self):
    v = ctypes.c_int64()
    self._chk(self._L.pcl_resident_completed(self._h, ctypes.byref(v)))
    return v.value



def attach():
    for f in (resident_start, resident_post, resident_wait, resident_eval, resident_stop, resident_completed):
        setattr(_integ._PclContext, f.__name__, f)

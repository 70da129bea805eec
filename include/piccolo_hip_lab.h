/* piccolo_hip_lab.h -- entry points that exist ONLY in lab builds of the library (hipcc ... -DPCL_LAB; piccolo.jl_amd/_lib.py:
 * build_library(lab=True) -> csrc/libpiccolo_hip_lab.so).  None of them is something the reference's FFI for this path would bind, and the shipped
 * libpiccolo_hip.so does not export them (tests/test_abi_cpu.py asserts that): a variant that was built, measured and lost, and two debugging aids.
 * They stay in the tree so that the measurements in DESIGN.md / lab/probes/README.md can be reproduced. */
#ifndef PICCOLO_HIP_LAB_H
#define PICCOLO_HIP_LAB_H
#include "piccolo_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* RESIDENT evaluator -- pcl_eval_jac_dev without a launch per evaluation.  What it replaces on the reference's side is still
 * evaluate! + eval_jacobian of the dynamics constraint (src/control/integrators.jl:620-640, 780-790; the solver calls them back to back
 * once per iteration): a solver iteration is a request, not a kernel launch.
 *   pcl_resident_start   kernel 4's workgroups (one per CU) go resident on a stream of their own and wait for requests; Z_dev, delta_dev
 *                        (may be NULL) and vals_dev are fixed until pcl_resident_stop -- the caller rewrites Z_dev IN PLACE between requests
 *                        (after pcl_resident_wait; copies and kernels on other streams are seen: the caches are invalidated per request)
 *   pcl_resident_post    `count` more evaluations of the CURRENT Z_dev (asynchronous; count > 1 only makes sense for measurements)
 *   pcl_resident_wait    returns when every posted evaluation is complete and its delta / vals are visible to the host, to copies and to kernels
 *                        of any stream (spins on a host word the device writes; timeout_s <= 0: none)
 *   pcl_resident_stop    the workgroups leave; the context is as before
 * delta and vals are bitwise those of pcl_eval_jac_dev (same code, compiled as a function).  The kernel leaves by itself after
 * option resident_idle_us (default 5000) without a request -- a blocked host never hangs the device -- and the next post starts it
 * again (get_option resident_launches counts the starts).  While it is resident other kernels find the CUs' LDS taken: launches of
 * this context on its own stream (pcl_hess_dev ...) run when a CU can hold them beside it or when it leaves; calls that synchronise the
 * DEVICE (hipMalloc, hipFree, hipDeviceSynchronize) wait for it to leave.  Needs what kernel 4 needs (PCL_ESHAPE otherwise).
 * MEASURED (MI355X, config 3, one trajectory, order 4): 29.5-32.6 us per evaluation with requests posted ahead against 24.2 for launches queued
 * ahead, 37.8-39.5 us per request round trip against 36 for launch + pcl_sync -- the hand-over of a request costs what a launch costs; the
 * entry points exist so that this can be reproduced, nothing takes them by default. */
int pcl_resident_start(pcl_ctx *ctx, const double *Z_dev, double *delta_dev, double *vals_dev);
int pcl_resident_post(pcl_ctx *ctx, int32_t count);
int pcl_resident_wait(pcl_ctx *ctx, double timeout_s);
int pcl_resident_stop(pcl_ctx *ctx);
int pcl_resident_stamps(pcl_ctx *ctx, int64_t *out, int64_t count); /* debugging: see piccolo_hip.hip */
int pcl_resident_completed(const pcl_ctx *ctx, int64_t *count); /* evaluations complete since pcl_resident_start (-1: never started) */

/* Profiling aid (lab builds with -DPCL_PROFILE as well): after pcl_set_option(ctx, "debug_timing", 1), up to 64
 * s_memtime stamps written by workgroup 0 at its phase boundaries during the last launch. */
int pcl_debug_timing(pcl_ctx *ctx, int64_t *out, int64_t cap);

/* The generator's term tables of the pattern-compiled kernels applied on the host to one column:
 * y = (G0[0] + sum_l u_l G_l) x, x and y of length n = 2d -- what the generated product computes, checkable without a GPU
 * (the CPU suite checks the same two functions of pcl_codegen_v4.hpp through a host-only shim: tests/test_abi_cpu.py). */
int pcl_codegen_apply_v4(int d, int m, const double *G0, int n_g0, const double *Gj, const double *u, const double *x, double *y, int transposed /* 1: y = G(u)^T x */);

#ifdef __cplusplus
}
#endif
#endif /* PICCOLO_HIP_LAB_H */

// piccolo_hip.hip -- libpiccolo_hip.so: C ABI (include/piccolo_hip.h cites the reference interfaces it replaces) and host
// side of the gfx950 (MI355X / CDNA4) Pade collocation constraint evaluator.  One translation unit; the kernels live in
//   pcl_device_common.hpp      parameter block, MFMA tile GEMM on LDS operands, wave-level helpers
//   pcl_kernel_fused_v3.hpp    default residual + Jacobian kernel: persistent, one workgroup per CU, stream / matrix roles
//   pcl_kernels_fused_v2.hpp   fallback (two workgroups per CU); also kets and the compact Jacobian at large n
//   pcl_kernels_reference.hpp  single-role kernel (A/B reference) and the general-order kernel (Pade 2..10)
//   pcl_kernel_eval.hpp        residual only (pcl_eval): persistent, three barriers per interval
//   pcl_kernels_hessian.hpp    Hessian of the Lagrangian: versions 1 (one workgroup per interval) and 2 (column chunks, fallback)
//   pcl_kernel_hessian_v3.hpp  Hessian of the Lagrangian, default: one workgroup per interval, jobs split by drive
//   pcl_kernels_misc.hpp       compact -> full expansion, rollout, derivative / time rows, terminal infidelity
// DESIGN.md has the full account.  No CPU fallback exists: every entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include "piccolo_hip.h"
#ifdef PCL_LAB
#include "piccolo_hip_lab.h"
#endif
#include "pcl_codegen.hpp"
#include "pcl_codegen_v4.hpp"

#define PCL_VERSION_STR "piccolo_hip 0.4.0 (gfx950, pade 2/4/6/8/10)"

#include "pcl_device_common.hpp"
#include "pcl_kernels_reference.hpp"
#include "pcl_kernel_pade_v2.hpp"
#include "pcl_kernels_fused_v2.hpp"
#include "pcl_kernel_fused_v3.hpp"
#include "pcl_kernel_eval.hpp"
#include "pcl_kernel_fused_small.hpp"
#include "pcl_kernels_hessian.hpp"
#include "pcl_kernel_hessian_v3.hpp"
#include "pcl_kernels_misc.hpp"
#include "pcl_kernels_objective.hpp"
#include "pcl_host_expand.hpp"

// ------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------
static thread_local std::string g_create_error;

struct pcl_ctx {
    pcl_desc desc;
    int n, K;
    int cols;  // state columns (d for unitaries, 1 for kets)
    int vec = 0;  // PCL_STATE_VECTOR: n = desc.d (general generator on one column; general-order kernel only)
    long long x_dim;
    std::vector<int32_t> x_offs;
    int device;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // device copies
    double *dG0 = nullptr, *ducoef = nullptr, *dcsr_val = nullptr, *dcsc_val = nullptr;
    double *dGjd = nullptr;  // small systems (n <= 8): the drives dense, [m][n*n] column-major (pcl_fused_small_kernel)
    int *dupos = nullptr, *dcsr_ptr = nullptr, *dcsr_col = nullptr, *dcsc_ptr = nullptr, *dcsc_row = nullptr, *dxoffs = nullptr;
    int n_upos = 0;
    int *dumap = nullptr, *dell_col = nullptr;
    double *dell_val = nullptr;
    int ell_w = 0, iso = 0, uell_w = 0;
    unsigned char *duell_l = nullptr;
    double *duell_v = nullptr;
    int *dellt_col = nullptr;  // ELL form of G_l^T (Hessian kernel v2)
    double *dellt_val = nullptr;
    int ellt_w = 0;
    int drives_antisym = 0;  // every G_l == -G_l^T exactly
    double *dug0 = nullptr;
    int64_t opt_general_threads = 512;
    int64_t opt_general_version = 0;   // 0 auto (the lock-step kernel where it fits) | 1 reference formulation | 2 lock-step kernel or error
    int64_t opt_general_slices = 0;    // lock-step kernel: slices per interval (0 auto)
    double *dreduce = nullptr;  // staging of pcl_reduce_sum (host buffer)
    int64_t reduce_cap = 0;
    double *dexpm = nullptr, *dxout = nullptr;  // rollout scratch: propagators, staged output of the host-pointer call
    double *dhpart = nullptr;  // Hessian v2 scratch: per (b,k,slice) partial scalar entries + per (b,k) arrival counters
    unsigned int *dhcnt = nullptr;
    long long hpart_cap = 0;
    int64_t opt_hess_kernel = 0, last_hess_kernel = 0;  // 0 = auto
    // pattern-compiled kernels (pcl_codegen.hpp): plan, device tables, per-interval value table of G(u_k)
    pcl_codegen::SpPlan *sp_plan = nullptr;
    int *dsp_pos = nullptr;
    double *dsp_coef = nullptr, *dsp_glv = nullptr, *dsp_gvals = nullptr;
    long long sp_gvals_cap = 0;  // intervals the value table holds
    int sp_failed = 0;           // the source did not compile: the other kernels serve the context
    int sp_hess_unfit = 0;       // the Hessian kernel's tiles do not fit LDS (d = 32 with 5 or 6 drives)
    hipFunction_t sp_fval = nullptr, sp_fhess = nullptr, sp_feval = nullptr;  // compiled on first use, kept for the context's lifetime
    int *dsp_pos_n = nullptr;       // the same tables in the emission order of the residual kernel's products
    double *dsp_coef_n = nullptr;
    // pattern-compiled FUSED residual + Jacobian kernel (pcl_codegen_v4.hpp, any Pade order): plan, drift tables, magnitudes
    pcl_codegen::V4Plan *v4_plan = nullptr;
    double *dv4_tab = nullptr, *dv4_tab_t = nullptr, *dv4_mags = nullptr, *dv4_dcf = nullptr;
    hipFunction_t v4_ft = nullptr;  // the fused kernel of the module WITH the slice-ticket roles (launches of several trajectories)
    int v4_ft_failed = 0;           // ... could not be had: those launches take the static split of v4_f (same bits)
    // v4_ticket = -1 (auto), launches of several trajectories: static split or slice tickets, decided PER VALUES ARRAY by timing both on it (the
    // two give the same bits).  The static split's time depends on where the array's pages live (181-228 us per 8 seeds by array and box), the
    // tickets' hardly (193-217): neither wins everywhere.  Per array: 2 untimed launches, then 3 + 3 timed ones alternating (events on the launch
    // stream, read back without blocking), then the faster variant for good (tickets unless the static split is 2 % ahead).
    struct V4Tune {
        const void *key = nullptr;
        long long units = 0;
        int calls = 0, choice = -1, done[2] = {0, 0}, pend_variant[8] = {0};
        float best[2] = {1e30f, 1e30f};
        hipEvent_t ev[8][2] = {};
        bool pend[8] = {false};
        unsigned long long stamp = 0;
    } v4_tune[4];
    unsigned long long v4_tune_clock = 0;
    int64_t last_v4_tune_choice = -1, last_v4_tune_static_ns = 0, last_v4_tune_ticket_ns = 0;
    hipFunction_t v4_f = nullptr, v4_feval = nullptr, v4_fevalc = nullptr /* cooperative residual kernel (optional) */, v4_fhess = nullptr, v4_fhess2 = nullptr /* two workgroups per interval */;
    hipFunction_t v4_fhessc = nullptr;  // general-order Hessian, one wave per group of state columns (pcl_kernel_hess_cols.hpp)
    double *dhcx = nullptr;             // ... the waves' rows of reduced sums and the intervals' arrival counters (self-resetting)
    unsigned int *dhcc = nullptr;
    long long hc_cap = 0;
    double *dhcr = nullptr;             // ... [interval][q - 2][d][n]
    unsigned int *dhcf = nullptr;       // ... [interval] R-chain waves that have delivered (self-resetting)
    long long hcr_cap = 0;
    int64_t opt_hess_rpre = -1;         // -1 auto (launches of more than n_cu / 2 intervals: 1) | 0 the chain inside the column-group waves | 1 R-chain waves in the same launch
    int64_t last_hess_rpre = 0;
    hipFunction_t v4_fhessp = nullptr;  // ... launches of at most n_cu / 2 intervals (one trajectory): a chain wave and a contribution wave per column group (pcl_hess_cols_pair_kernel)
    int64_t opt_hess_pair = -1;         // -1 auto (on for such launches) | 0 one wave per column group | 1 wherever the kernel fits
    int64_t last_hess_pair = 0;
    int v4_hessc_failed = 0;
    double *dh4x = nullptr;        // general-order Hessian, two workgroups per interval: their rows of reduced sums ...
    unsigned int *dh4c = nullptr;  // ... and the arrival counters (self-resetting)
    long long h4_cap = 0;
    int64_t opt_hess_split = -1, last_hess_split = 0;  // -1 auto (launches of at most n_cu / 2 intervals) | 0 | 1
    int64_t opt_v4_tune = 1;  // v4_ticket auto: 1 the static split or the slice tickets by timing both on the values array | 0 always tickets (round 4)
    int64_t opt_hess_xcd = -1;  // column-group Hessian kernel: the waves of an interval on one XCD (-1 auto: 8 | 0 / 1 blockIdx order | n)
    int64_t opt_eval_coop = -1, last_eval_coop = 0;  // residual only: four waves per interval (-1 auto: launches of at most two intervals per CU)
    int v4_hess_failed = 0;
    int v4_failed = 0;
    int64_t opt_v4_variant = 0;     // PCL_PROFILE builds: timing variants of the generated product (wrong results)
    int last_objective_launches = 0;  // what the last pcl_objective_dev launched: 1 (one grid) | 2
    int64_t opt_objective_launches = 0;  // 0 auto | 2: always the two launches (A/B, tests)
    int last_step_launches = 0;  // what the last pcl_eval_jac_merit_objective_dev launched (get_option)
    int64_t opt_v4_flags = 0, opt_v4_np = 0;  // kernel 4 A/B switches (KParams::v4_flags); tiles of the powers of G (0 auto)
    int64_t opt_v4_ticket = -1;     // kernel 4: slice tickets (-1 auto: full-value launches of several intervals per CU at orders 2 and 4 | 0 static split | 1)
    int64_t opt_v4_ticket_cols = 0; // ... state columns per slice ticket (0 auto: 3)
    int64_t opt_v4_ticket_ahead = 2; // ... when the next slice is asked for: 0 when the stream waves have issued this one's stores | 1 a slice ahead | 2 at this one's last column (default)
    int64_t opt_v4_group = 0;       // ... workgroups per group (0 auto: 8, one per XCD)
    bool ticket_launched = false;   // a launch that works on the context's self-resetting device counters (slice tickets; the arrival counters and exchange rows of the split / column-group Hessian kernels) has been enqueued on the current stream (see change_stream)
    int64_t last_v4_ticket = 0;     // state columns per block ticket of the last kernel-4 launch (0: static work split)
    unsigned int *dv4_tick = nullptr;  // ... [block ticket, pipelines gone, chain ticket]: zero between launches (the last pipeline out resets them)
    int *herr = nullptr, *derr = nullptr;  // device error word (host-mapped): a barrier-free kernel whose bounded wait gave up sets bit 0
#ifdef PCL_LAB
    // RESIDENT evaluator (pcl_resident_*; pcl_kernel_fused_sparse.hpp, SP4_RESIDENT): kernel 4's workgroups stay on the device and run one
    // evaluation per posted request
    struct Resident {
        hipStream_t stream = nullptr;  // its own: work queued behind a resident kernel waits for it to leave
        hipFunction_t f = nullptr;
        unsigned *hbox = nullptr, *hbox_dev = nullptr, *dbox = nullptr;  // the host's words (mapped) and the device's (layout: see the kernel)
        unsigned *hinit = nullptr;                                        // pinned source of the device words' start values
        KParams *dparams = nullptr, *hparams = nullptr;                   // the evaluation's parameter block in device memory (the kernel reads it per request, where it is used) and its pinned source
        KParams p;
        const double *tab = nullptr, *dcf = nullptr;
        long long grid = 0;
        size_t lds = 0;
        unsigned block = 0, posted = 0;
        bool active = false, launched = false;
        int64_t launches = 0;
    } res;
    bool res_capture = false;            // launch_fused_v4 fills `res` instead of launching
    int64_t opt_resident_idle_us = 5000; // the resident kernel leaves after this long without a request (the next request starts it again)
#else
    static constexpr bool res_capture = false;  // (the shipped library has no resident evaluator: include/piccolo_hip_lab.h)
#endif
    int64_t opt_v4_tail_mode = 3;   // kernel 4: who stores delta and the tails: 0 the writer wave | 1 ... nontemporal | 2 ... write-through | 3 the stream waves (default)
    int64_t opt_eval_kernel = 0;    // 0 auto | 1 matrix-core residual kernel | 2 pattern-compiled
    // staging for the host-pointer entry points
    double *dZ = nullptr, *dmu = nullptr, *ddelta = nullptr, *dvals = nullptr, *dhess = nullptr;
    // options
    // host-pointer entry points: pinned staging + compact D2H + threaded expansion into the caller's array
    pcl_host::Pool *pool = nullptr;
    double *hZ = nullptr, *hcompact = nullptr, *hdelta = nullptr;  // pinned (hipHostMalloc)
    double *dcomp_host = nullptr;                                    // device buffer of the compact values
    hipEvent_t ev_chunk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t opt_prof = 0;  // -DPCL_PROFILE builds only
    int64_t opt_host_threads = 0, opt_host_path = 0, opt_host_chunks = 4;
    int64_t opt_host_store_bytes = 0, last_host_store_bytes = 0;  // streaming-store width of the host expansion: 0 the widest the host has | 16 | 32 | 64
    int host_threads_tuned = 0;       // auto: the thread count the sweep over the context's first host-delivered calls found fastest
    int host_tune_calls = 0;          // ... calls sampled so far
    double host_tune_t[8] = {1e300, 1e300, 1e300, 1e300, 1e300, 1e300, 1e300, 1e300};
    double host_expand_GBps = 0.0;    // ... the delivered rate of the fastest sampled call
    int win_first = 0, win_count = 0;  // member window (pcl_set_member_window): the members / seeds the evaluator entry points cover
    int64_t opt_cols_per_slice = 0, opt_use_mfma = 1, opt_nt = -1 /* auto by launch size */, opt_kernel = 0;  // 0 = auto: 3 where its specialised instance applies, else 1 / 2 by shape
    long long *ddbg = nullptr;
    void *comm = nullptr;  // ncclComm_t
    double *dgoal = nullptr;  // iso-vec of the goal unitary (pcl_set_goal) or of its subspace block (pcl_set_goal_subspace)
    int *dsub = nullptr;      // subspace indices of an embedded goal
    int n_sub = 0;            // 0: full-space fidelity
    double *dweights = nullptr;  // per member / seed weights of the objective (NULL: ones)
    // the terminal loss in its general form F = c'x + sum_r (A_r'x)^2 (pcl_kernels_objective.hpp): set by pcl_set_goal_form (form_user: value and
    // gradient go through it too) or derived from the unitary goal (for the Hessian of the objective only)
    double *dformA = nullptr, *dformc = nullptr, *dgram = nullptr, *dcoef = nullptr;
    int form_R = 0, form_L = 0, form_scope = 0;
    bool form_user = false, gram_ready = false;
    std::vector<PclReg> regs;    // quadratic regularisers (pcl_add_regularizer)
    std::vector<double> reg_R;
    PclReg *dregs = nullptr;
    double *dreg_R = nullptr;
    bool regs_dirty = false;
    double *dobj = nullptr;      // objective scratch: per-member values | per-knot regulariser values
    double *dphik = nullptr;     // merit scratch: per-interval partial sums
    double *dmcols = nullptr;    // fused merit: per-column partial dot products written by fused kernel 3
    unsigned int *dmticket = nullptr;  // ... arrival ticket of pcl_merit_finish_kernel (zero between launches)
    bool tickets_dirty = true;   // the arrival tickets (objective sum, merit finish) are re-zeroed on the stream before their next use:
                                 // set at allocation and whenever the stream changes (the last-arriving workgroup resets them otherwise)
    const double *merit_lam = nullptr;  // set for the duration of pcl_eval_jac_merit_dev
    int merit_want = 0, merit_fused = 0;
    double *dgrad = nullptr, *dval = nullptr;  // staging of the host-pointer objective call
    int64_t opt_specialize = 1;
    int64_t opt_jit = 1;      // compile shape-specialised instances on first use (hiprtc) for shapes outside the static table
    int64_t opt_require_jit = 0;  // 1: a pattern-compiled kernel that cannot be had (no libhiprtc, no headers, compile error) is an error, not a fallback
    int64_t jit_fallbacks = 0;    // pattern-compiled kernels this context wanted and did not get (the matrix-core / general kernels serve it)
    int64_t opt_general = 0;  // 1: run the general-order kernel also for pade_order 4 (cross-check)
    int64_t opt_contig = -1;     // v3: contiguous column ranges per workgroup (-1: auto by launch size)
    int64_t opt_stream_wg = -1;  // v3, contiguous: stream-role workgroups (-1: auto = half, 0: every workgroup does both)
    int64_t last_n_stream = 0;  // stream-role workgroups of the last kernel-3 launch (0: fused roles / round-robin)
    int64_t last_kernel = 0;  // 10*version + (1 if shape-specialised) of the last fused launch
    int64_t opt_grid = 0;  // 0: resident workgroups (persistent kernel)
    size_t lds_set[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const void *lds_kern[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // last MaxDynamicSharedMemorySize set per kernel variant
    int max_lds = 0;
    int n_cu = 0;
    // order policy (pade_order = 0 in the descriptor): host copies of the generators for the norm bound, the tolerance, what was found
    std::vector<double> hG0, hGj;
    double order_tol = 1e-10, order_theta = 0.0;
    int order_tol_met = 1;  // 0: the order policy had to settle for order 10 above its tolerance (note_order)
    mutable std::string err;
};

static int fail(const pcl_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ctx, PCL_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Every entry point runs on the context's device and leaves the caller's current device as it found it.
namespace {
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            changed = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};
}  // namespace
#define ON_DEVICE(ctx)                                                                                         \
    DeviceGuard dev_guard_((ctx)->device);                                                                     \
    if (dev_guard_.err != hipSuccess) return fail(ctx, PCL_EHIP, "hipSetDevice(%d): %s", (ctx)->device, hipGetErrorString(dev_guard_.err))

// LDS of the fused pattern-compiled kernel: m + 4 chain tiles (D, S, W, V, dW_l) + np tiles of the powers of G + counters
static size_t v4_lds_bytes(int d, int m, int np) { return ((size_t)(m + 4 + np) * d * (2 * d + 1) + 24) * sizeof(double); }
// tiles for the powers of G: q + 1 when they fit (one-item launches use q: the stream never waits for the P wave; ticket launches let the
// P wave start the next item's first power while the stream folds this item's last), else as many as fit (>= 2)
static int v4_power_tiles(int d, int m, int q, size_t max_lds) {
    int np = q + 1;
    while (np > 2 && v4_lds_bytes(d, m, np) > max_lds) --np;
    return v4_lds_bytes(d, m, np) <= max_lds ? np : 0;
}

static long long jac_per_full(const pcl_ctx *c) {
    return 2LL * c->cols * c->n * c->n + c->x_dim * (c->desc.n_drives + 1);
}
static long long jac_per_compact(const pcl_ctx *c) { return 2LL * c->n * c->n + c->x_dim * (c->desc.n_drives + 1); }
static long long hess_per(const pcl_ctx *c) {
    const long long m = c->desc.n_drives;
    return (m + 1) * (m + 2) / 2 + 2 * c->x_dim * (m + 1);
}
static long long z_len(const pcl_ctx *c) {
    return (long long)c->desc.z_dim * c->desc.N * (c->desc.batch_mode == PCL_BATCH_TRAJ ? c->desc.batch : 1);
}
static long long n_rows(const pcl_ctx *c) { return (long long)c->win_count * c->x_dim * c->K; }        // of the member window
static long long n_rows_all(const pcl_ctx *c) { return (long long)c->desc.batch * c->x_dim * c->K; }  // of every member (allocation sizes)

extern "C" const char *pcl_version(void) { return PCL_VERSION_STR; }

extern "C" const char *pcl_last_error(const pcl_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

template <class T>
static int upload(pcl_ctx *ctx, T **dst, const std::vector<T> &src) {
    const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIP_TRY(ctx, hipMalloc((void **)dst, bytes));
    if (!src.empty()) HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return PCL_OK;
}

extern "C" int pcl_create(const pcl_desc *dsc, pcl_ctx **out) {
    if (!out) return fail(nullptr, PCL_EINVAL, "pcl_create: out is NULL");
    *out = nullptr;
    if (!dsc) return fail(nullptr, PCL_EINVAL, "pcl_create: desc is NULL");
    if (dsc->struct_size != (int32_t)sizeof(pcl_desc))
        return fail(nullptr, PCL_EINVAL, "pcl_create: desc.struct_size=%d, library expects %zu (ABI mismatch)",
                    dsc->struct_size, sizeof(pcl_desc));
    const bool vec = dsc->state_cols == PCL_STATE_VECTOR;  // general real d x d generator on one real column
    const int d = dsc->d, m = dsc->n_drives, n = vec ? d : 2 * d;
    if (d < 1 || m < 0 || dsc->N < 2 || dsc->batch < 1)
        return fail(nullptr, PCL_EINVAL, "pcl_create: need d>=1, n_drives>=0, N>=2, batch>=1 (got d=%d m=%d N=%d batch=%d)", d,
                    m, dsc->N, dsc->batch);
    if (n > 2 * PCL_MAX_D)
        return fail(nullptr, PCL_ESHAPE, "pcl_create: generator dimension %d exceeds %d (LDS-resident tiles; d <= %d)", n, 2 * PCL_MAX_D, PCL_MAX_D);
    if (m > 24) return fail(nullptr, PCL_ESHAPE, "pcl_create: n_drives=%d exceeds 24", m);
    if (dsc->pade_order != 0 && dsc->pade_order != 2 && dsc->pade_order != 4 && dsc->pade_order != 6 && dsc->pade_order != 8 && dsc->pade_order != 10)
        return fail(nullptr, PCL_ENOTIMPL, "pcl_create: pade_order=%d; diagonal Pade orders 2, 4, 6, 8, 10 are implemented (0: chosen by pcl_set_order_policy)", dsc->pade_order);
    if (dsc->index_base != 0 && dsc->index_base != 1) return fail(nullptr, PCL_EINVAL, "pcl_create: index_base must be 0 or 1");
    if (dsc->batch_mode != PCL_BATCH_MEMBERS && dsc->batch_mode != PCL_BATCH_TRAJ)
        return fail(nullptr, PCL_EINVAL, "pcl_create: unknown batch_mode %d", dsc->batch_mode);
    if (!dsc->G0 || (m > 0 && !dsc->Gj) || !dsc->x_offs) return fail(nullptr, PCL_EINVAL, "pcl_create: G0/Gj/x_offs must be non-NULL");
    if (dsc->state_cols < 0 && !vec) return fail(nullptr, PCL_EINVAL, "pcl_create: state_cols=%d", dsc->state_cols);
    const int cols = vec ? 1 : (dsc->state_cols > 0 ? dsc->state_cols : d);
    if (cols > d) return fail(nullptr, PCL_EINVAL, "pcl_create: state_cols=%d exceeds d=%d", cols, d);
    const long long x_dim = (long long)n * cols;
    const int n_off = dsc->batch_mode == PCL_BATCH_MEMBERS ? dsc->batch : 1;
    for (int i = 0; i < n_off; ++i)
        if (dsc->x_offs[i] < 0 || dsc->x_offs[i] + x_dim > dsc->z_dim)
            return fail(nullptr, PCL_EINVAL, "pcl_create: x_offs[%d]=%d with x_dim=%lld does not fit z_dim=%d", i, dsc->x_offs[i],
                        x_dim, dsc->z_dim);
    if (dsc->u_off < 0 || dsc->u_off + m > dsc->z_dim || dsc->dt_off < 0 || dsc->dt_off >= dsc->z_dim)
        return fail(nullptr, PCL_EINVAL, "pcl_create: u_off/dt_off outside the knot (z_dim=%d)", dsc->z_dim);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PCL_EHIP, "pcl_create: no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (dsc->device_id < 0 || dsc->device_id >= ndev) return fail(nullptr, PCL_EINVAL, "pcl_create: device_id %d of %d", dsc->device_id, ndev);

    pcl_ctx *ctx = new (std::nothrow) pcl_ctx();
    if (!ctx) return fail(nullptr, PCL_ENOMEM, "pcl_create: out of host memory");
    ctx->desc = *dsc;
    ctx->n = n;
    ctx->K = dsc->N - 1;
    ctx->x_dim = x_dim;
    ctx->cols = cols;
    ctx->vec = vec ? 1 : 0;
    ctx->x_offs.assign(dsc->x_offs, dsc->x_offs + n_off);
    ctx->hG0.assign(dsc->G0, dsc->G0 + (size_t)n * n * (dsc->per_member_G0 ? dsc->batch : 1));
    if (m > 0) ctx->hGj.assign(dsc->Gj, dsc->Gj + (size_t)m * n * n);
    ctx->desc.x_offs = nullptr;
    ctx->desc.G0 = ctx->desc.Gj = nullptr;
    ctx->device = dsc->device_id;
    ctx->win_first = 0;
    ctx->win_count = dsc->batch;
    if (const char *tk = getenv("PCL_V4_TICKET")) {  // default of option "v4_ticket" (A/B of whole programs: -1 auto | 0 static split | 1 slice tickets)
        const long v = strtol(tk, nullptr, 10);
        if (v >= -1 && v <= 1) ctx->opt_v4_ticket = v;
    }
    if (const char *hp = getenv("PCL_HOST_PATH")) {  // default of option "host_path" (1: full values over PCIe, 2: compact + host expansion)
        const long v = strtol(hp, nullptr, 10);
        if (v >= 0 && v <= 2) ctx->opt_host_path = v;
    }

#define CREATE_TRY(expr)                                                  \
    do {                                                                  \
        int rc_ = (expr);                                                 \
        if (rc_ != PCL_OK) {                                              \
            g_create_error = ctx->err;                                    \
            pcl_destroy(ctx);                                             \
            return rc_;                                                   \
        }                                                                 \
    } while (0)
#define CREATE_HIP(expr)                                                                               \
    do {                                                                                               \
        hipError_t e2_ = (expr);                                                                       \
        if (e2_ != hipSuccess) {                                                                       \
            fail(nullptr, PCL_EHIP, "pcl_create: %s: %s", #expr, hipGetErrorString(e2_));              \
            pcl_destroy(ctx);                                                                          \
            return PCL_EHIP;                                                                           \
        }                                                                                              \
    } while (0)

    DeviceGuard dev_guard_(ctx->device);
    CREATE_HIP(dev_guard_.err);
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, PCL_EHIP, "pcl_create: device %d is %s; this library is built for gfx950 only", ctx->device, prop.gcnArchName);
        pcl_destroy(ctx);
        return PCL_EHIP;
    }
    ctx->max_lds = (int)prop.maxSharedMemoryPerMultiProcessor;
    ctx->n_cu = prop.multiProcessorCount;
    CREATE_HIP(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;

    // --- drive structures: union pattern (+ coefficient table), CSR and CSC of every G_l ------
    const size_t nn = (size_t)n * n;
    std::vector<int> upos;
    std::vector<double> ucoef;
    for (size_t pz = 0; pz < nn; ++pz) {
        bool any = false;
        for (int l = 0; l < m; ++l) any |= dsc->Gj[l * nn + pz] != 0.0;
        if (any) {
            upos.push_back((int)pz);
            for (int l = 0; l < m; ++l) ucoef.push_back(dsc->Gj[l * nn + pz]);
        }
    }
    ctx->n_upos = (int)upos.size();
    std::vector<int> csr_ptr((size_t)std::max(m, 1) * (n + 1), 0), csr_col, csc_ptr((size_t)std::max(m, 1) * (n + 1), 0), csc_row;
    std::vector<double> csr_val, csc_val;
    for (int l = 0; l < m; ++l) {
        const double *A = dsc->Gj + l * nn;  // column-major: A[i + n*j]
        for (int i = 0; i < n; ++i) {
            csr_ptr[(size_t)l * (n + 1) + i] = (int)csr_col.size();
            for (int j = 0; j < n; ++j)
                if (A[i + (size_t)n * j] != 0.0) {
                    csr_col.push_back(j);
                    csr_val.push_back(A[i + (size_t)n * j]);
                }
        }
        csr_ptr[(size_t)l * (n + 1) + n] = (int)csr_col.size();
        for (int j = 0; j < n; ++j) {
            csc_ptr[(size_t)l * (n + 1) + j] = (int)csc_row.size();
            for (int i = 0; i < n; ++i)
                if (A[i + (size_t)n * j] != 0.0) {
                    csc_row.push_back(i);
                    csc_val.push_back(A[i + (size_t)n * j]);
                }
        }
        csc_ptr[(size_t)l * (n + 1) + n] = (int)csc_row.size();
    }
    int uell_w = 0;
    for (size_t q = 0; q < upos.size(); ++q) {
        int cnt = 0;
        for (int l = 0; l < m; ++l) cnt += ucoef[q * m + l] != 0.0;
        uell_w = std::max(uell_w, cnt);
    }
    uell_w = std::max(uell_w, 1);
    std::vector<unsigned char> uell_l(std::max<size_t>(upos.size(), 1) * uell_w, 0);
    std::vector<double> uell_v(std::max<size_t>(upos.size(), 1) * uell_w, 0.0);
    for (size_t q = 0; q < upos.size(); ++q) {
        int cnt = 0;
        for (int l = 0; l < m; ++l)
            if (ucoef[q * m + l] != 0.0) {
                uell_l[q * uell_w + cnt] = (unsigned char)l;
                uell_v[q * uell_w + cnt] = ucoef[q * m + l];
                ++cnt;
            }
    }
    ctx->uell_w = uell_w;
    std::vector<int> umap(nn, -1);
    for (size_t q = 0; q < upos.size(); ++q) umap[upos[q]] = (int)q;
    int ell_w = m > 0 ? 1 : 0;
    for (int l = 0; l < m; ++l)
        for (int i = 0; i < n; ++i) ell_w = std::max(ell_w, csr_ptr[(size_t)l * (n + 1) + i + 1] - csr_ptr[(size_t)l * (n + 1) + i]);
    std::vector<int> ell_col((size_t)m * n * ell_w, 0);
    std::vector<double> ell_val((size_t)m * n * ell_w, 0.0);
    for (int l = 0; l < m; ++l)
        for (int i = 0; i < n; ++i) {
            const int beg = csr_ptr[(size_t)l * (n + 1) + i], end = csr_ptr[(size_t)l * (n + 1) + i + 1];
            for (int q = beg; q < end; ++q) {
                ell_col[((size_t)l * n + i) * ell_w + (q - beg)] = csr_col[q];
                ell_val[((size_t)l * n + i) * ell_w + (q - beg)] = csr_val[q];
            }
        }
    ctx->ell_w = ell_w;
    int ellt_w = m > 0 ? 1 : 0;
    for (int l = 0; l < m; ++l)
        for (int i = 0; i < n; ++i) ellt_w = std::max(ellt_w, csc_ptr[(size_t)l * (n + 1) + i + 1] - csc_ptr[(size_t)l * (n + 1) + i]);
    std::vector<int> ellt_col((size_t)m * n * ellt_w, 0);
    std::vector<double> ellt_val((size_t)m * n * ellt_w, 0.0);
    for (int l = 0; l < m; ++l)
        for (int i = 0; i < n; ++i) {
            const int beg = csc_ptr[(size_t)l * (n + 1) + i], end = csc_ptr[(size_t)l * (n + 1) + i + 1];
            for (int q = beg; q < end; ++q) {
                ellt_col[((size_t)l * n + i) * ellt_w + (q - beg)] = csc_row[q];
                ellt_val[((size_t)l * n + i) * ellt_w + (q - beg)] = csc_val[q];
            }
        }
    ctx->ellt_w = ellt_w;
    {
        bool anti = true;
        for (int l = 0; l < m && anti; ++l)
            for (int j = 0; j < n && anti; ++j)
                for (int i = 0; i < n; ++i)
                    if (dsc->Gj[(size_t)l * nn + i + (size_t)n * j] != -dsc->Gj[(size_t)l * nn + j + (size_t)n * i]) {
                        anti = false;
                        break;
                    }
        ctx->drives_antisym = anti ? 1 : 0;
    }
    // exact iso structure  M = [[A, -B], [B, A]]  of the drift(s) and of every drive?
    auto is_iso = [&](const double *A) {
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < d; ++i) {
                if (A[(i + d) + (size_t)n * (j + d)] != A[i + (size_t)n * j]) return false;
                if (A[i + (size_t)n * (j + d)] != -A[(i + d) + (size_t)n * j]) return false;
            }
        return true;
    };
    bool iso = !vec;  // the iso(.) block structure only exists for n = 2d
    if (iso)
        for (int bb = 0; bb < (dsc->per_member_G0 ? dsc->batch : 1); ++bb) iso = iso && is_iso(dsc->G0 + bb * nn);
    if (iso)
        for (int l = 0; l < m; ++l) iso = iso && is_iso(dsc->Gj + l * nn);
    ctx->iso = iso ? 1 : 0;
    // pattern-compiled kernels: sparse iso generators of a unitary problem, one state column per lane (d <= 32), m + 2 waves
    if (iso && cols == d && d >= 9 && d <= 32 && m >= 1 && m <= 6) {
        pcl_codegen::SpPlan plan = pcl_codegen::make_plan(d, m, dsc->G0, dsc->per_member_G0 ? dsc->batch : 1, dsc->Gj);
        if (plan.ok && plan.nz <= 640 && (double)plan.nz <= 0.45 * 2.0 * d * d) ctx->sp_plan = new pcl_codegen::SpPlan(std::move(plan));
    }
    if (ctx->sp_plan) {  // the fused kernel of the same family: one LDS tile per chain (m + 7 tiles of d (n + 1) doubles)
        pcl_codegen::V4Plan v4 = pcl_codegen::make_v4_plan(d, m, dsc->G0, dsc->per_member_G0 ? dsc->batch : 1, dsc->Gj);
        if (v4.ok && v4_power_tiles(d, m, std::max(dsc->pade_order, 2) / 2, (size_t)ctx->max_lds) > 0) ctx->v4_plan = new pcl_codegen::V4Plan(std::move(v4));
    }
    std::vector<double> g0(dsc->G0, dsc->G0 + nn * (dsc->per_member_G0 ? dsc->batch : 1));
    CREATE_TRY(upload(ctx, &ctx->dG0, g0));
    if (n <= 16 && m >= 1) {
        std::vector<double> gjd(dsc->Gj, dsc->Gj + nn * m);
        CREATE_TRY(upload(ctx, &ctx->dGjd, gjd));
    }
    CREATE_TRY(upload(ctx, &ctx->dupos, upos));
    CREATE_TRY(upload(ctx, &ctx->ducoef, ucoef));
    CREATE_TRY(upload(ctx, &ctx->dcsr_ptr, csr_ptr));
    CREATE_TRY(upload(ctx, &ctx->dcsr_col, csr_col));
    CREATE_TRY(upload(ctx, &ctx->dcsr_val, csr_val));
    CREATE_TRY(upload(ctx, &ctx->dcsc_ptr, csc_ptr));
    CREATE_TRY(upload(ctx, &ctx->dcsc_row, csc_row));
    CREATE_TRY(upload(ctx, &ctx->dcsc_val, csc_val));
    CREATE_TRY(upload(ctx, &ctx->dumap, umap));
    CREATE_TRY(upload(ctx, &ctx->duell_l, uell_l));
    CREATE_TRY(upload(ctx, &ctx->duell_v, uell_v));
    CREATE_TRY(upload(ctx, &ctx->dell_col, ell_col));
    CREATE_TRY(upload(ctx, &ctx->dell_val, ell_val));
    CREATE_TRY(upload(ctx, &ctx->dellt_col, ellt_col));
    CREATE_TRY(upload(ctx, &ctx->dellt_val, ellt_val));
    {
        std::vector<double> ug0(std::max<size_t>(upos.size(), 1), 0.0);
        for (size_t q = 0; q < upos.size(); ++q) ug0[q] = dsc->G0[upos[q]];
        CREATE_TRY(upload(ctx, &ctx->dug0, ug0));
    }
    std::vector<int> xo(ctx->x_offs.begin(), ctx->x_offs.end());
    CREATE_TRY(upload(ctx, &ctx->dxoffs, xo));
    if (ctx->sp_plan) {
        const pcl_codegen::SpPlan &sp = *ctx->sp_plan;
        std::vector<double> glv(sp.mags);  // the distinct magnitudes of the drives' entries
        glv.resize(std::max<size_t>(glv.size(), 1) + 16, 0.0);
        CREATE_TRY(upload(ctx, &ctx->dsp_pos, sp.pos));
        CREATE_TRY(upload(ctx, &ctx->dsp_coef, sp.coef));
        CREATE_TRY(upload(ctx, &ctx->dsp_glv, glv));
        CREATE_TRY(upload(ctx, &ctx->dsp_pos_n, sp.pos_n));
        CREATE_TRY(upload(ctx, &ctx->dsp_coef_n, sp.coef_n));
    }
    if (ctx->v4_plan) {
        const pcl_codegen::V4Plan &v4 = *ctx->v4_plan;
        const int nb = dsc->per_member_G0 ? dsc->batch : 1;
        std::vector<double> tab((size_t)nb * v4.n_drift_pad, 0.0);
        for (int bb = 0; bb < nb; ++bb)
            for (size_t q = 0; q < v4.drift_pos.size(); ++q) tab[(size_t)bb * v4.n_drift_pad + q] = dsc->G0[(size_t)bb * nn + v4.drift_pos[q]];
        std::vector<double> mg(v4.mags);
        mg.resize(std::max<size_t>(mg.size(), 1) + 16, 0.0);
        std::vector<double> tab_t((size_t)nb * v4.n_drift_pad, 0.0);  // the order G(u)^T x reads the streamed drift entries in
        for (int bb = 0; bb < nb; ++bb)
            for (size_t q = 0; q < v4.drift_pos_t.size(); ++q) tab_t[(size_t)bb * v4.n_drift_pad + q] = dsc->G0[(size_t)bb * nn + v4.drift_pos_t[q]];
        CREATE_TRY(upload(ctx, &ctx->dv4_tab, tab));
        CREATE_TRY(upload(ctx, &ctx->dv4_tab_t, tab_t));
        CREATE_TRY(upload(ctx, &ctx->dv4_mags, mg));
        CREATE_TRY(upload(ctx, &ctx->dv4_dcf, v4.dcf_vals));
        const size_t tick_bytes = (4 + (size_t)dsc->batch * (dsc->N - 1)) * sizeof(unsigned int);
        CREATE_HIP(hipMalloc((void **)&ctx->dv4_tick, tick_bytes));
        CREATE_HIP(hipMemset(ctx->dv4_tick, 0, tick_bytes));
    }
    // device error word, host-mapped: a kernel whose bounded wait gave up sets it; every evaluator entry point and pcl_sync look at it
    CREATE_HIP(hipHostMalloc((void **)&ctx->herr, 64, hipHostMallocMapped));
    *ctx->herr = 0;
    CREATE_HIP(hipHostGetDevicePointer((void **)&ctx->derr, ctx->herr, 0));
#undef CREATE_TRY
#undef CREATE_HIP
    *out = ctx;
    return PCL_OK;
}

extern "C" int pcl_comm_destroy(pcl_ctx *ctx);
extern "C" void pcl_destroy(pcl_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard dev_guard_(ctx->device);
#ifdef PCL_LAB
    if (ctx->res.active) (void)pcl_resident_stop(ctx);  // before any buffer it reads on each request is freed
#endif
    (void)pcl_comm_destroy(ctx);
    void *ptrs[] = {ctx->dhcr, ctx->dhcf, ctx->dhcx, ctx->dhcc, ctx->dh4x, ctx->dh4c, ctx->dGjd, ctx->dG0, ctx->ducoef, ctx->dcsr_val, ctx->dcsc_val, ctx->dupos, ctx->dcsr_ptr, ctx->dcsr_col,
                    ctx->dcsc_ptr, ctx->dcsc_row, ctx->dxoffs, ctx->dZ, ctx->dmu, ctx->ddelta, ctx->dvals, ctx->dhess,
                    ctx->dumap, ctx->dell_col, ctx->dell_val, ctx->duell_l, ctx->duell_v, ctx->ddbg, ctx->dellt_col, ctx->dellt_val,
                    ctx->dhpart, ctx->dhcnt, ctx->dug0, ctx->dexpm, ctx->dxout, ctx->dreduce};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (ctx->dgoal) (void)hipFree(ctx->dgoal);
    delete ctx->pool;
    ctx->pool = nullptr;
    for (double *q : {ctx->hZ, ctx->hcompact, ctx->hdelta})
        if (q) (void)hipHostFree(q);
    if (ctx->dcomp_host) (void)hipFree(ctx->dcomp_host);
    for (hipEvent_t e : ctx->ev_chunk)
        if (e) (void)hipEventDestroy(e);
    for (auto &t : ctx->v4_tune)
        for (auto &pr : t.ev)
            for (hipEvent_t e : pr)
                if (e) (void)hipEventDestroy(e);
    for (void *q : {(void *)ctx->dformA, (void *)ctx->dformc, (void *)ctx->dgram, (void *)ctx->dcoef})
        if (q) (void)hipFree(q);
    for (void *q : {(void *)ctx->dsub, (void *)ctx->dweights, (void *)ctx->dregs, (void *)ctx->dreg_R, (void *)ctx->dobj, (void *)ctx->dphik, (void *)ctx->dmcols, (void *)ctx->dmticket,
                    (void *)ctx->dgrad, (void *)ctx->dval})
        if (q) (void)hipFree(q);
    for (void *q : {(void *)ctx->dsp_pos, (void *)ctx->dsp_coef, (void *)ctx->dsp_glv, (void *)ctx->dsp_gvals, (void *)ctx->dsp_pos_n, (void *)ctx->dsp_coef_n})
        if (q) (void)hipFree(q);
    delete ctx->sp_plan;
    for (void *q : {(void *)ctx->dv4_tab, (void *)ctx->dv4_tab_t, (void *)ctx->dv4_mags, (void *)ctx->dv4_dcf, (void *)ctx->dv4_tick})
        if (q) (void)hipFree(q);
#ifdef PCL_LAB
    if (ctx->res.stream) (void)hipStreamDestroy(ctx->res.stream);
    if (ctx->res.hbox) (void)hipHostFree(ctx->res.hbox);
    if (ctx->res.hinit) (void)hipHostFree(ctx->res.hinit);
    if (ctx->res.hparams) (void)hipHostFree(ctx->res.hparams);
    if (ctx->res.dparams) (void)hipFree(ctx->res.dparams);
    if (ctx->res.dbox) (void)hipFree(ctx->res.dbox);
#endif
    if (ctx->herr) (void)hipHostFree(ctx->herr);
    delete ctx->v4_plan;
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int pcl_constraint_dim(const pcl_ctx *ctx, int64_t *x_dim, int64_t *rows, int64_t *cols) {
    if (!ctx) return PCL_EINVAL;
    if (x_dim) *x_dim = ctx->x_dim;
    if (rows) *rows = n_rows(ctx);
    if (cols) *cols = z_len(ctx) + ctx->desc.global_dim;
    return PCL_OK;
}
extern "C" int pcl_jac_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *per) {
    if (!ctx) return PCL_EINVAL;
    if (per) *per = jac_per_full(ctx);
    if (nnz) *nnz = jac_per_full(ctx) * ctx->win_count * ctx->K;
    return PCL_OK;
}
extern "C" int pcl_set_member_window(pcl_ctx *ctx, int32_t first, int32_t count) {
    if (!ctx) return PCL_EINVAL;
    if (first < 0 || count < 1 || (long long)first + count > ctx->desc.batch)
        return fail(ctx, PCL_EINVAL, "pcl_set_member_window: [%d, %d) outside the %d members of this context", first, first + count, ctx->desc.batch);
    ctx->win_first = first;
    ctx->win_count = count;
    return PCL_OK;
}
extern "C" int pcl_jac_compact_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *per) {
    if (!ctx) return PCL_EINVAL;
    if (per) *per = jac_per_compact(ctx);
    if (nnz) *nnz = jac_per_compact(ctx) * ctx->win_count * ctx->K;
    return PCL_OK;
}
extern "C" int pcl_hess_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *per) {
    if (!ctx) return PCL_EINVAL;
    if (per) *per = hess_per(ctx);
    if (nnz) *nnz = hess_per(ctx) * ctx->win_count * ctx->K;
    return PCL_OK;
}

template <class I>
static int jac_structure_impl(const pcl_ctx *ctx, I *rows, I *cols) {
    if (!ctx) return PCL_EINVAL;
    if (!rows || !cols) return fail(ctx, PCL_EINVAL, "pcl_jac_structure: NULL output");
    const pcl_desc &D = ctx->desc;
    const long long n = ctx->n, d = ctx->cols, m = D.n_drives, xd = ctx->x_dim, zd = D.z_dim, base = D.index_base;
    const long long per = jac_per_full(ctx);
    for (long long b = 0; b < ctx->win_count; ++b) {  // b: member inside the window, bg: member of the context
        const long long bg = ctx->win_first + b;
        const long long xo = ctx->x_offs[D.batch_mode == PCL_BATCH_MEMBERS ? bg : 0];
        const long long voff = D.batch_mode == PCL_BATCH_TRAJ ? bg * zd * D.N : 0;
        for (long long k = 0; k < ctx->K; ++k) {
            I *r = rows + (b * ctx->K + k) * per, *c = cols + (b * ctx->K + k) * per;
            const long long r0 = b * xd * ctx->K + k * xd + base;
            long long p = 0;
            for (int seg = 0; seg < 2; ++seg) {
                const long long cb = voff + (k + seg) * zd + xo + base;
                for (long long cc = 0; cc < d; ++cc)
                    for (long long j = 0; j < n; ++j)
                        for (long long i = 0; i < n; ++i, ++p) {
                            r[p] = (I)(r0 + cc * n + i);
                            c[p] = (I)(cb + cc * n + j);
                        }
            }
            for (long long cc = 0; cc < d; ++cc)  // tail: per state column, the m drive blocks then the dt block
                for (long long l = 0; l <= m; ++l) {
                    const long long col = voff + k * zd + (l < m ? D.u_off + l : D.dt_off) + base;
                    for (long long i = 0; i < n; ++i, ++p) {
                        r[p] = (I)(r0 + cc * n + i);
                        c[p] = (I)col;
                    }
                }
        }
    }
    return PCL_OK;
}
extern "C" int pcl_jac_structure(const pcl_ctx *ctx, int32_t *rows, int32_t *cols) {
    if (ctx && (n_rows(ctx) + 1 > INT32_MAX || z_len(ctx) + ctx->desc.global_dim + 1 > INT32_MAX))
        return fail(ctx, PCL_ESHAPE, "pcl_jac_structure: indices exceed int32; use pcl_jac_structure_i64");
    return jac_structure_impl<int32_t>(ctx, rows, cols);
}
extern "C" int pcl_jac_structure_i64(const pcl_ctx *ctx, int64_t *rows, int64_t *cols) {
    return jac_structure_impl<int64_t>(ctx, rows, cols);
}

template <class I>
static int hess_structure_impl(const pcl_ctx *ctx, I *rows, I *cols) {
    if (!ctx) return PCL_EINVAL;
    if (!rows || !cols) return fail(ctx, PCL_EINVAL, "pcl_hess_structure: NULL output");
    const pcl_desc &D = ctx->desc;
    const long long m = D.n_drives, xd = ctx->x_dim, zd = D.z_dim, base = D.index_base;
    const long long per = hess_per(ctx);
    for (long long b = 0; b < ctx->win_count; ++b) {
        const long long bg = ctx->win_first + b;
        const long long xo = ctx->x_offs[D.batch_mode == PCL_BATCH_MEMBERS ? bg : 0];
        const long long voff = D.batch_mode == PCL_BATCH_TRAJ ? bg * zd * D.N : 0;
        for (long long k = 0; k < ctx->K; ++k) {
            I *r = rows + (b * ctx->K + k) * per, *c = cols + (b * ctx->K + k) * per;
            const long long uk = voff + k * zd + D.u_off, hk = voff + k * zd + D.dt_off;
            const long long xk = voff + k * zd + xo, xn = voff + (k + 1) * zd + xo;
            long long p = 0;
            auto put = [&](long long a, long long bb) {
                r[p] = (I)(std::max(a, bb) + base);
                c[p] = (I)(std::min(a, bb) + base);
                ++p;
            };
            for (long long i = 0; i < m; ++i)
                for (long long j = 0; j <= i; ++j) put(uk + i, uk + j);
            for (long long j = 0; j < m; ++j) put(hk, uk + j);
            put(hk, hk);
            for (long long l = 0; l < m; ++l)
                for (long long q = 0; q < xd; ++q) put(uk + l, xk + q);
            for (long long q = 0; q < xd; ++q) put(hk, xk + q);
            for (long long l = 0; l < m; ++l)
                for (long long q = 0; q < xd; ++q) put(xn + q, uk + l);
            for (long long q = 0; q < xd; ++q) put(xn + q, hk);
        }
    }
    return PCL_OK;
}
extern "C" int pcl_hess_structure(const pcl_ctx *ctx, int32_t *rows, int32_t *cols) {
    if (ctx && z_len(ctx) + ctx->desc.global_dim + 1 > INT32_MAX)
        return fail(ctx, PCL_ESHAPE, "pcl_hess_structure: indices exceed int32; use pcl_hess_structure_i64");
    return hess_structure_impl<int32_t>(ctx, rows, cols);
}
extern "C" int pcl_hess_structure_i64(const pcl_ctx *ctx, int64_t *rows, int64_t *cols) {
    return hess_structure_impl<int64_t>(ctx, rows, cols);
}

#include "pcl_host_jit.hpp"
// --- launch helpers -------------------------------------------------------------------------
// LD = (n rounded up to 4) + 2  ==  2*odd: conflict-free ds_read_b64 of the MFMA b operand
// (16 columns x 2 k-rows per half-wave land on 32 distinct 8-byte bank pairs).
static int lds_ld(int n) { return ((n + 3) & ~3) + 2; }  // n = generator dimension

static void fill_params(const pcl_ctx *ctx, KParams &p) {
    memset(&p, 0, sizeof p);
    const pcl_desc &D = ctx->desc;
    p.G0 = ctx->dG0 + (D.per_member_G0 ? (long long)ctx->win_first * ctx->n * ctx->n : 0);  // member window: offsets, not kernel logic
    p.upos = ctx->dupos;
    p.ucoef = ctx->ducoef;
    p.n_upos = ctx->n_upos;
    p.csr_ptr = ctx->dcsr_ptr;
    p.csr_col = ctx->dcsr_col;
    p.csr_val = ctx->dcsr_val;
    p.csc_ptr = ctx->dcsc_ptr;
    p.csc_row = ctx->dcsc_row;
    p.csc_val = ctx->dcsc_val;
    p.x_offs = ctx->dxoffs + (D.batch_mode == PCL_BATCH_MEMBERS ? ctx->win_first : 0);
    p.x_off0 = (D.batch_mode != PCL_BATCH_MEMBERS || ctx->win_count == 1) ? (int)ctx->x_offs[D.batch_mode == PCL_BATCH_MEMBERS ? ctx->win_first : 0] : -1;
    p.umap = ctx->dumap;
    p.uell_l = ctx->duell_l;
    p.uell_v = ctx->duell_v;
    p.uell_w = ctx->uell_w;
    p.ell_val = ctx->dell_val;
    p.ell_col = ctx->dell_col;
    p.ell_w = ctx->ell_w;
    p.ellt_val = ctx->dellt_val;
    p.ellt_col = ctx->dellt_col;
    p.ellt_w = ctx->ellt_w;
    p.hpart = ctx->dhpart;
    p.hcnt = ctx->dhcnt;
    p.ug0 = ctx->dug0;
    p.iso = ctx->iso;
    p.z_batch_stride = D.batch_mode == PCL_BATCH_TRAJ ? (long long)D.z_dim * D.N : 0;
    p.g0_batch_stride = D.per_member_G0 ? (long long)ctx->n * ctx->n : 0;
    p.d = D.d;
    p.cols = ctx->cols;
    p.n = ctx->n;
    p.m = D.n_drives;
    p.K = ctx->K;
    p.z_dim = D.z_dim;
    p.u_off = D.u_off;
    p.dt_off = D.dt_off;
    p.batch = ctx->win_count;
    p.LD = lds_ld(ctx->n);
    p.nt = ctx->opt_nt > 0 ? (int)ctx->opt_nt : 0;  // (auto: the fused launch decides by its size)
    p.dbg = ctx->ddbg;
    p.prof = (int)ctx->opt_prof;
    p.hess_per = hess_per(ctx);
    p.err = ctx->derr;
}

// A pattern-compiled kernel was wanted (jit = 1, the shape applies) and could not be had: the slower kernel families serve the context
// (1.5-2.2x for residual + Jacobian, 25-30x for the Hessian at orders 6-10) -- never silently: the note is what pcl_last_error returns
// until the next failure, "jit_fallbacks" counts, and with option require_jit = 1 the call fails instead.
static int jit_fell_back(pcl_ctx *ctx, const char *what) {
    ++ctx->jit_fallbacks;
    {
        std::lock_guard<std::mutex> lock(g_jit_mutex);
        ++g_jit_fallbacks;
    }
    fail(ctx, PCL_EHIP, "%s: the pattern-compiled kernel is not available (%s); %s", what, g_jit_note.c_str(),
         ctx->opt_require_jit ? "require_jit = 1" : "falling back to the built-in kernels (slower; set option require_jit = 1 to make this an error)");
    if (getenv("PCL_VERBOSE")) fprintf(stderr, "piccolo_hip: %s\n", ctx->err.c_str());
    return ctx->opt_require_jit ? PCL_EHIP : PCL_ENOTIMPL;
}

// A barrier-free kernel whose bounded wait gave up (a logic or timing failure: its outputs are partly stale) has set the context's
// error word.  Looked at by every evaluator entry point before it launches and by every call that synchronises.
static int check_device_error(pcl_ctx *ctx, const char *where) {
    if (!ctx->herr) return PCL_OK;
    const int w = __atomic_load_n(ctx->herr, __ATOMIC_ACQUIRE);
    if (!w) return PCL_OK;
    __atomic_store_n(ctx->herr, 0, __ATOMIC_RELEASE);
    if (ctx->dv4_tick) (void)hipMemsetAsync(ctx->dv4_tick, 0, (4 + (size_t)ctx->desc.batch * ctx->K) * sizeof(unsigned int), ctx->stream);  // the launch may have left its counters behind
    // ... and the self-resetting arrival counters of the Hessian kernels (a wave that gave up never made its arrival: the interval's counter -- and the R-chain
    // waves' delivery word, whose stale 1 would pass the NEXT launch the old tiles -- would stay where they are)
    if (ctx->dhcc && ctx->hc_cap > 0) (void)hipMemsetAsync(ctx->dhcc, 0, (size_t)ctx->hc_cap * sizeof(unsigned int), ctx->stream);
    if (ctx->dhcf) (void)hipMemsetAsync(ctx->dhcf, 0, (size_t)ctx->desc.batch * ctx->K * sizeof(unsigned int), ctx->stream);
    if (ctx->dh4c && ctx->h4_cap > 0) (void)hipMemsetAsync(ctx->dh4c, 0, (size_t)ctx->h4_cap * sizeof(unsigned int), ctx->stream);
    return fail(ctx, PCL_EINTERNAL, "%s: an earlier kernel of this context gave up a bounded wait between its waves (device error word 0x%x); "
                "the outputs of that launch are incomplete", where, w);
}

static size_t fused_lds_bytes(const KParams &p, bool jac) {  // version-1 kernel
    const size_t ncols1 = jac ? (size_t)(2 + p.m) * p.nc : 2 * (size_t)p.nc;
    size_t dbl = (size_t)p.LD * p.n * (jac ? 2 : 1) + 2 * p.LD * ncols1 + 2 * (size_t)p.LD * p.nc + 8 + p.m;
    return dbl * sizeof(double);
}

static const size_t ELL_LDS_MAX_BYTES = 8192;

static size_t fused2_lds_bytes(const KParams &p, bool jac, bool ell_lds) {  // version-2 kernel
    const size_t ncols1 = jac ? (size_t)(2 + p.m) * p.nc : 2 * (size_t)p.nc;
    const size_t n_ell = (size_t)p.m * p.n * p.ell_w;
    size_t bytes = ((size_t)p.LD * p.n * (jac ? 2 : 1) + 2 * p.LD * ncols1 + (size_t)p.LD * p.nc + 2 * (p.m + 1)) * sizeof(double);
    if (jac && ell_lds) bytes += n_ell * sizeof(double) + (n_ell * sizeof(unsigned short) + 7) / 8 * 8;
    return bytes + 128;  // slack: operand tiles may be read past the last buffer's edge
}

static bool ell_fits_lds(const pcl_ctx *ctx) {
    const size_t n_ell = (size_t)ctx->desc.n_drives * ctx->n * ctx->ell_w;
    return n_ell > 0 && n_ell * (sizeof(double) + sizeof(unsigned short)) <= ELL_LDS_MAX_BYTES;
}

// State columns per workgroup.  Every slice recomputes G(u_k)^2, so fewer, wider slices do less
// arithmetic; but (i) two workgroups must fit in one CU's LDS so that one streams while the other
// computes, and (ii) the grid has to cover the chip a few times over.
static int choose_cols_per_slice(const pcl_ctx *ctx, bool jac) {
    const int d = ctx->cols;
    if (ctx->opt_cols_per_slice > 0) return (int)std::min<int64_t>(ctx->opt_cols_per_slice, d);
    KParams p;
    memset(&p, 0, sizeof p);
    p.n = ctx->n;
    p.m = ctx->desc.n_drives;
    p.LD = ((ctx->n + 3) & ~3) + 2;
    p.ell_w = ctx->ell_w;
    const bool v2 = ctx->opt_kernel == 0 || ctx->opt_kernel >= 2;
    const bool ell = ell_fits_lds(ctx);
    auto bytes = [&](int nc) {
        p.nc = nc;
        return v2 ? fused2_lds_bytes(p, jac, ell) : fused_lds_bytes(p, jac);
    };
    const long long bk = (long long)ctx->win_count * ctx->K;
    const long long want = 3LL * std::max(ctx->n_cu, 1);
    int best = 1;
    for (int nc = d; nc >= 1; --nc) {
        if (bytes(nc) > (size_t)ctx->max_lds / 2 && nc > 1) continue;  // keep two workgroups per CU
        const long long S = (d + nc - 1) / nc;
        best = nc;
        if (!jac || bk * S >= want) break;
    }
    return best;
}

static bool v3_supported(const pcl_ctx *ctx) { return 2 + ctx->desc.n_drives <= 16; }
static int v3_ncw(const pcl_ctx *ctx, int nc) { return std::max(1, std::min(nc, 16 / (2 + ctx->desc.n_drives))); }

static size_t fused3_lds_bytes(const pcl_ctx *ctx, const KParams &p, bool tab) {  // version-3 kernel
    const size_t tile = (size_t)p.LD * p.n, wsz = (size_t)p.LD * (16 + 3 * p.ncw);
    const size_t n_ell = (size_t)p.m * p.n * p.ell_w, n_un = (size_t)ctx->n_upos, uw = (size_t)ctx->uell_w;
    size_t bytes = (4 * tile + 4 * wsz + 3 * (size_t)(p.m + 1) + 2) * sizeof(double);
    if (tab) bytes += (n_un * uw + n_un + n_ell) * sizeof(double) + (n_un + n_ell) * 2 + n_un * uw;
    return (bytes + 7) / 8 * 8 + 128;  // slack: operand tiles may be read past the last buffer's edge
}

// v3 cost model: one workgroup per CU walks ceil(items / CUs) items; an item costs max(store stream, matrix work).
// Constants are the measured config-3 phase times (scripts/phase_timing.py, with the stream running): ~7 us per
// 2-column chunk, ~4 us for G(u) + G^2, stream at ~0.85 of the CU's fair HBM share; other shapes scale by MFMA count.
// Role split needs the chunk buffers of matrix waves 4..7 inside the second halves of the G / G^2 double buffers.
// the shape of BASELINE configs 3/4/5 (three 3-level transmons: d = 27, six drives with two entries per row)
// plus two 5-level transmons (d = 25, four drives) and, for launches of one round of workgroups, two 4-level transmons
static bool v3_role_split_fits(const pcl_ctx *ctx);
static bool hess_v2_supported(const pcl_ctx *ctx);
static bool v3_specialised(const pcl_ctx *ctx) {
    if (!ctx->opt_specialize || ctx->ell_w < 1 || ctx->ell_w > 2) return false;
    const int d = ctx->desc.d, m = ctx->desc.n_drives;
    if (ctx->ell_w == 2 && ((d == 27 && m == 6) || (d == 25 && m == 4))) return true;
    if (ctx->ell_w == 2 && d == 16 && m == 4 && (long long)ctx->win_count * ctx->K <= 512) return true;
    // Other shapes: `kernel_version = 3` compiles the shape on first use (hiprtc) instead of running the run-time-shape
    // instance; measured on d = 22..30 it only ties the persistent two-workgroup kernels, so `auto` does not take it.
    return false;
}
static bool v3_role_split_fits(const pcl_ctx *ctx) {
    const int ncw = v3_ncw(ctx, ctx->desc.d), LD = lds_ld(ctx->n);
    return 2 * (size_t)LD * (16 + 3 * ncw) <= (size_t)LD * ctx->n;
}
// Work split of kernel 3: contiguous column ranges (+ role split) pay off once every CU has a few intervals' worth of
// columns; below that the round-robin slices balance a short launch better (measured: batch >= 3 at config 3).
static bool v3_contiguous(const pcl_ctx *ctx) {
    if (ctx->opt_cols_per_slice > 0 || ctx->opt_contig == 0) return false;
    if (ctx->opt_contig > 0) return true;
    const long long cols = (long long)ctx->win_count * ctx->K * ctx->desc.d;
    return v3_role_split_fits(ctx) && cols >= 28LL * std::max(ctx->n_cu, 1);
}
static int choose_cols_v3(const pcl_ctx *ctx) {
    const int d = ctx->desc.d, n = ctx->n, m = ctx->desc.n_drives;
    if (ctx->opt_cols_per_slice > 0) return (int)std::min<int64_t>(ctx->opt_cols_per_slice, d);
    const double hbm = 0.85 * 6.3e12;  // store-stream rate the kernel sustains chip-wide
    const long long bk = (long long)ctx->win_count * ctx->K;
    const int rt = (n + 15) / 16, ks = (n + 3) / 4;
    const int g2ct = ctx->iso ? (d + 15) / 16 : (n + 15) / 16;
    const double t_chunk = 7.0e-6 * (rt * ks) / (4.0 * 14.0);
    const double t_build = 4.0e-6 * (((rt + 3) / 4) * ((g2ct + 1) / 2) * ks) / 14.0;
    std::vector<double> tt(d + 1, 1e300);
    double best_t = 1e300;
    for (int nc = 1; nc <= d; ++nc) {
        const int ncw = v3_ncw(ctx, nc);
        const long long S = (d + nc - 1) / nc;
        const long long items = bk * S, ncu = std::max(ctx->n_cu, 1);
        const int chunks_per_wave = ((nc + ncw - 1) / ncw + 3) / 4;
        const double t_matrix = t_build + chunks_per_wave * t_chunk;
        const double item_bytes = 2.0 * nc * n * n * 8.0;
        // rounds of up to n_cu items; the HBM rate is shared by the workgroups active in the round (a lone CU tops out
        // at a few times its fair share)
        double t = 0.0;
        for (long long left = items; left > 0; left -= ncu) {
            const double active = (double)std::min(left, ncu);
            const double rate = std::min(hbm / active, 3.0 * hbm / (double)ncu);
            t += std::max(item_bytes / rate, t_matrix);
        }
        // a ragged last slice (d % nc != 0) leaves workgroups with unequal items: charge the mean fill
        const double fill = (double)d / (double)(S * nc);
        tt[nc] = t / std::sqrt(std::max(fill, 0.25)) + 2.0 * t_build;
        best_t = std::min(best_t, tt[nc]);
    }
    // among near-ties take the narrowest slice (more, smaller items balance better across the CUs)
    int best = d;
    for (int nc = d; nc >= 1; --nc)
        if (tt[nc] <= 1.06 * best_t) best = nc;
    return best;
}

static int set_lds_attr(pcl_ctx *ctx, const void *kern, int slot, size_t lds) {
    if (ctx->lds_set[slot] == lds && ctx->lds_kern[slot] == kern) return PCL_OK;
    ctx->lds_kern[slot] = kern;
    HIP_TRY(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->lds_set[slot] = lds;
    return PCL_OK;
}

// General-order kernel (pade_order != 4, or option general_pade_kernel): slice width from the LDS budget.
static size_t pade_lds_bytes(const KParams &p, bool jac) {
    const size_t tiles = (jac ? 3 : 1) * (size_t)p.LD * p.n;
    const size_t percol = (size_t)(p.q + 1) + 2 + (jac ? 2 + 2 * (size_t)p.m : 0);
    return (tiles + percol * p.LD * p.nc + 8 + p.m) * sizeof(double);
}
// Lock-step general-order kernel (pcl_kernel_pade_v2.hpp).  PCL_ENOTIMPL (no error text): the shape is not taken, the caller
// falls back to the reference formulation.
static int launch_pade_v2(pcl_ctx *ctx, KParams &p) {
    const int n = p.n, m = p.m, cols = p.cols;
    if ((n & 1) || n > 64 || n < 2 || !p.jac) return PCL_ENOTIMPL;
    const int LD = n | 1;
    auto lds_of = [&](int S) {
        const int nc = (cols + S - 1) / S, npc = (n + S - 1) / S;
        return ((size_t)((2 + 2 * (2 + m)) * nc + n + npc) * LD + 80 + m + 8) * sizeof(double);
    };
    const long long items = (long long)p.batch * p.K;
    const int s_max = std::max(cols, 1);
    int S = 1;
    while (S < s_max && lds_of(S) > (size_t)ctx->max_lds) ++S;
    if (lds_of(S) > (size_t)ctx->max_lds) return PCL_ENOTIMPL;
    if (ctx->opt_general_slices > 0)
        S = (int)std::max<int64_t>(S, std::min<int64_t>(ctx->opt_general_slices, s_max));
    else  // few intervals: more, narrower slices while the grid still fits the CUs in one round
        S = std::max(S, (int)std::min<long long>(s_max, ctx->n_cu / std::max(items, 1LL)));
    p.LD = LD;
    p.nc = (cols + S - 1) / S;
    p.S = S;  // (slices beyond the last state column still carry their columns of the powers)
    size_t lds = lds_of(S);
    const size_t ell_bytes = (size_t)m * n * p.ell_w * (sizeof(double) + sizeof(int)) + 16;
    p.ell_lds = m > 0 && lds + ell_bytes <= (size_t)ctx->max_lds;
    if (p.ell_lds) lds += ell_bytes;
    p.lds_doubles = (int)(lds / sizeof(double));
    const long long grid = items * p.S;
    if (grid > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
    int rc = set_lds_attr(ctx, (const void *)pcl_pade_v2_kernel, 7, lds);
    if (rc != PCL_OK) return rc;
    hipLaunchKernelGGL(pcl_pade_v2_kernel, dim3((unsigned)grid), dim3(PV2_NT), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_kernel = 190 + p.q;
    ctx->last_n_stream = 0;
    return PCL_OK;
}

// diagonal Pade coefficients of exp at order 2q: c_j = (2q - j)! q! / ((2q)! j! (q - j)!)
static void fill_pade(KParams &p, int order) {
    p.q = order / 2;
    double f[16];
    f[0] = 1.0;
    for (int i = 1; i < 16; ++i) f[i] = f[i - 1] * i;
    for (int j = 0; j <= p.q; ++j) p.pc[j] = f[2 * p.q - j] * f[p.q] / (f[2 * p.q] * f[j] * f[p.q - j]);
}

// Pattern-compiled fused residual + Jacobian kernel (pcl_kernel_fused_sparse.hpp; any Pade order): sparse exact-iso generators of
// a unitary problem whose m + 7 chain tiles fit LDS.  PCL_ENOTIMPL (no error text): not taken, the caller goes on to the others.
static bool v4_available(const pcl_ctx *ctx) {
    return ctx->v4_plan && !ctx->v4_failed && ctx->opt_jit && !ctx->vec && ctx->cols == ctx->desc.d;
}
// the generated module of the context's system and order (both kernels), compiled on first use
static int v4_module(pcl_ctx *ctx, int q, int np) {
    if (ctx->v4_f && ctx->v4_feval) return PCL_OK;
    const std::string src = v4_source(*ctx->v4_plan, q, np, (int)ctx->opt_v4_variant);
    const std::string key = "fused-sparse:" + std::to_string(q) + ":" + std::to_string(std::hash<std::string>{}(src));
    ctx->v4_f = jit_compile(ctx->device, key, src, "pcl_fused_sparse_kernel", true);
    ctx->v4_feval = ctx->v4_f ? jit_compile(ctx->device, key, src, "pcl_eval_sparse4_kernel", true) : nullptr;
    ctx->v4_fevalc = (ctx->v4_feval && src.find("#define SP4_COOP 1") != std::string::npos) ? jit_compile(ctx->device, key, src, "pcl_eval_sparse4c_kernel", true) : nullptr;
    if (!ctx->v4_f || !ctx->v4_feval) {
        ctx->v4_f = ctx->v4_feval = ctx->v4_fevalc = nullptr;
        ctx->v4_failed = 1;
        return jit_fell_back(ctx, "residual + Jacobian");
    }
    return PCL_OK;
}
// Residual only on the same products: one wave per interval, 4 waves per workgroup (three tiles each)
static int launch_eval_v4(pcl_ctx *ctx, KParams &p) {
    if (!v4_available(ctx) || !p.delta) return PCL_ENOTIMPL;
    const pcl_codegen::V4Plan &v4 = *ctx->v4_plan;
    fill_pade(p, ctx->desc.pade_order);
    const int np = v4_power_tiles(p.d, p.m, p.q, (size_t)ctx->max_lds);
    if (!np) return PCL_ENOTIMPL;
    if (int rc = v4_module(ctx, p.q, np)) return rc;  // (PCL_ENOTIMPL: the caller goes on to the other kernels)
    const long long items = (long long)p.batch * p.K;
    if (items > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
    const int nw = 4;
    const size_t lds = (size_t)nw * 3 * p.d * (p.n + 1) * sizeof(double);
    const int per_cu = std::max(1, (int)((size_t)ctx->max_lds / lds));
    const long long slots = (long long)per_cu * std::max(ctx->n_cu, 1);
    long long grid = std::min<long long>((items + nw - 1) / nw, slots);
    if (ctx->opt_grid > 0) grid = std::min<long long>(ctx->opt_grid, (items + nw - 1) / nw);
    const double *tab = ctx->dv4_tab + (ctx->desc.per_member_G0 ? (long long)ctx->win_first * v4.n_drift_pad : 0);
    const double *dcf = ctx->dv4_dcf + (ctx->desc.per_member_G0 ? (long long)ctx->win_first * v4.n_dcf_pad : 0);
    void *args[] = {(void *)&p, (void *)&tab, (void *)&ctx->dv4_mags, (void *)&dcf};
    // small launches (up to two intervals per CU: a line-search trial on one to four trajectories): four waves per interval, the products in
    // four row ranges -- config 3, one trajectory: 8.7 -> 6.3 us per launch at order 4, 11.9 -> 7.8 at order 8; four trajectories 9.1 -> 7.9;
    // eight: 10.0 -> 11.8, so one wave per interval above (option eval_coop: -1 auto | 0 | 1)
    const bool coop = ctx->v4_fevalc && (ctx->opt_eval_coop == 1 || (ctx->opt_eval_coop < 0 && items <= 2LL * std::max(ctx->n_cu, 1)));
    ctx->last_eval_coop = coop ? 1 : 0;
    if (coop) {
        const size_t ldsc = (size_t)4 * p.d * (p.n + 1) * sizeof(double);
        const long long gc = ctx->opt_grid > 0 ? std::min<long long>(ctx->opt_grid, items) : std::min<long long>(items, 3LL * std::max(ctx->n_cu, 1));
        HIP_TRY(ctx, hipModuleLaunchKernel(ctx->v4_fevalc, (unsigned)gc, 1, 1, 64 * pcl_codegen::v4_parts(v4, p.q), 1, 1, (unsigned)ldsc, ctx->stream, args, nullptr));
    } else
        HIP_TRY(ctx, hipModuleLaunchKernel(ctx->v4_feval, (unsigned)grid, 1, 1, 64 * nw, 1, 1, (unsigned)lds, ctx->stream, args, nullptr));
    ctx->last_kernel = 80 + p.q;
    ctx->last_n_stream = 0;
    return PCL_OK;
}
static int launch_fused_v4(pcl_ctx *ctx, KParams &p, bool compact, bool want_merit = false);
// (the slice-ticket module could not be had: the same launch with the static split, from a fresh parameter block)
static int launch_fused_v4_static(pcl_ctx *ctx, KParams &p, bool compact, bool want_merit) {
    const int64_t keep = ctx->opt_v4_ticket;
    ctx->opt_v4_ticket = 0;
    p.tick = nullptr;
    p.tick_cpi = p.tick_G = p.tick_ahead = 0;
    const int rc = launch_fused_v4(ctx, p, compact, want_merit);
    ctx->opt_v4_ticket = keep;
    return rc;
}
static int launch_fused_v4(pcl_ctx *ctx, KParams &p, bool compact, bool want_merit) {
    if (!v4_available(ctx) || !p.jac) return PCL_ENOTIMPL;
    const pcl_codegen::V4Plan &v4 = *ctx->v4_plan;
    fill_pade(p, ctx->desc.pade_order);
    const int np = v4_power_tiles(p.d, p.m, p.q, (size_t)ctx->max_lds);
    if (!np) return PCL_ENOTIMPL;
    if (int rc = v4_module(ctx, p.q, np)) return rc;
    const int d = p.d, m = p.m;
    const long long bk = (long long)p.batch * p.K, ncu = std::max(ctx->n_cu, 1);
    // contiguous column ranges once every CU has about an interval's worth of columns; below that round-robin slices of the
    // intervals (an explicit cols_per_slice or contiguous = 0 / 1 decides otherwise)
    // (compact launches -- unique tiles only, the chains are all of the work -- deal the intervals round-robin: a contiguous range that
    //  ends inside an interval runs that interval's chains in two workgroups; 62.2 against 66.8 us per 8 trajectories)
    p.contig = ctx->opt_cols_per_slice > 0 ? 0 : (ctx->opt_contig >= 0 ? (int)ctx->opt_contig : (!compact && bk * d >= 28 * ncu ? 1 : 0));
    // SLICE TICKETS instead of a static split (launches of several trajectories; see the head of pcl_kernel_fused_sparse.hpp): groups of
    // workgroups walk the intervals in a static order and take slices of a few state columns from the interval's own counter, each when
    // the previous slice's stores are issued -- the front of addresses being written stays tight and the workgroups the memory side
    // serves first take more slices: the time no longer depends on where the values array's pages live.
    const int tick_G = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_v4_group > 0 ? ctx->opt_v4_group : 8, ncu));
    const bool ticket_ok = !ctx->res_capture && !compact && ctx->dv4_tick && !ctx->v4_ft_failed && ctx->opt_grid <= 0 && m + 10 <= 16 && ncu % tick_G == 0 &&
                           v4_lds_bytes(d, m, np) + 8 * 8 * 128 <= (size_t)ctx->max_lds;
    const bool ticket_auto = ticket_ok && ctx->opt_v4_ticket < 0 && p.q <= 2 && p.contig && ctx->opt_contig < 0 && ctx->opt_cols_per_slice <= 0;
    // auto: which of the two this values array gets (see v4_tune); timed launches are bracketed by events below
    pcl_ctx::V4Tune *tune = nullptr;
    int tune_slot = -1, tune_variant = 1;
    // (a stream that is being captured into a graph takes no part in the sampling: event records would become graph nodes and a query is illegal
    //  there; the launch gets the array's decided variant, else the tickets, and nothing of the sampling state moves)
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    const bool capturing = ticket_auto && ctx->opt_v4_tune != 0 && (hipStreamIsCapturing(ctx->stream, &cap_status) != hipSuccess || cap_status != hipStreamCaptureStatusNone);
    if (capturing) {
        (void)hipGetLastError();
        for (auto &t : ctx->v4_tune)
            if (t.key == (const void *)p.jac && t.units == bk && t.choice >= 0) tune_variant = t.choice;
    } else if (ticket_auto && ctx->opt_v4_tune != 0) {
        pcl_ctx::V4Tune *T = nullptr;
        for (auto &t : ctx->v4_tune)
            if (t.key == (const void *)p.jac && t.units == bk) T = &t;
        if (!T) {  // the least recently used entry makes room
            T = &ctx->v4_tune[0];
            for (auto &t : ctx->v4_tune)
                if (t.stamp < T->stamp) T = &t;
            for (int i = 0; i < 8; ++i) T->pend[i] = false;
            T->key = (const void *)p.jac, T->units = bk, T->calls = 0, T->choice = -1, T->done[0] = T->done[1] = 0, T->best[0] = T->best[1] = 1e30f;
        }
        T->stamp = ++ctx->v4_tune_clock;
        if (T->choice < 0) {
            for (int i = 0; i < 8; ++i)  // timings that have come back
                if (T->pend[i] && hipEventQuery(T->ev[i][1]) == hipSuccess) {
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, T->ev[i][0], T->ev[i][1]) == hipSuccess && ms > 0.f) {
                        T->best[T->pend_variant[i]] = std::min(T->best[T->pend_variant[i]], ms);
                        ++T->done[T->pend_variant[i]];
                    }
                    T->pend[i] = false;
                }
            (void)hipGetLastError();  // (hipErrorNotReady of a query is not an error of this call)
            if (T->done[0] >= 3 && T->done[1] >= 3) {
                T->choice = (T->best[0] * 1.02f < T->best[1]) ? 0 : 1;
                ctx->last_v4_tune_static_ns = (int64_t)(T->best[0] * 1e6f), ctx->last_v4_tune_ticket_ns = (int64_t)(T->best[1] * 1e6f);  // (ns)
            } else if (T->calls >= 40)
                T->choice = 1;  // (timings that never come back: tickets)
        }
        if (T->choice >= 0)
            tune_variant = T->choice;
        else {
            tune_variant = T->calls < 2 ? 1 : (T->calls & 1);
            if (T->calls >= 2 && T->done[tune_variant] < 3)
                for (int i = 0; i < 8 && tune_slot < 0; ++i)
                    if (!T->pend[i]) tune_slot = i;
            ++T->calls;
            tune = T;
        }
        ctx->last_v4_tune_choice = T->choice;
    }
    const bool ticket = ticket_ok && (ctx->opt_v4_ticket == 1 || (ticket_auto && tune_variant == 1));
    // (orders 6-10: the P wave's q products per visit are the longer chain -- 8 trajectories at order 8: 246-254 against 232 us)
    if (ticket) {
        // (slices of 4 state columns: 201.7 against 203.7 us per 8 seeds and 1.500 against 1.508 ms per 64 for slices of 3, the round-4 default; 5 the same, 2 / 7 / 9 slower: round 5)
        p.tick_cpi = ctx->opt_v4_ticket_cols > 0 ? (int)std::min<int64_t>(ctx->opt_v4_ticket_cols, d) : std::min(4, d);
        while ((d + p.tick_cpi - 1) / p.tick_cpi > 31) ++p.tick_cpi;  // (the slice index travels in five bits)
        p.tick = ctx->dv4_tick;
        p.tick_G = tick_G;
        p.tick_ahead = (int)ctx->opt_v4_ticket_ahead;
        p.contig = 0;
    }
    ctx->last_v4_ticket = ticket ? p.tick_cpi : 0;
    if (p.contig)
        p.nc = d;
    else if (ctx->opt_cols_per_slice > 0)
        p.nc = (int)std::min<int64_t>(ctx->opt_cols_per_slice, d);
    else {
        const long long S = std::max<long long>(1, std::min<long long>(d, ncu / std::max<long long>(bk, 1)));
        p.nc = (int)((d + S - 1) / S);
    }
    p.S = (d + p.nc - 1) / p.nc;
    p.all_matrix = 0;
    p.n_stream = 0;
    p.tail_mode = (int)ctx->opt_v4_tail_mode;
    p.v4_flags = (int)ctx->opt_v4_flags;
    // write-through block stores (auto, launches that fit the infinity cache) pay at orders 8 and 10 only: one trajectory, plain against
    // write-through: 24.9 / 25.4 us at order 2, 27.2 / 27.6 at 4, 28.1 / 28.9 at 6, 32.4 / 31.9 at 8.  Orders 2 and 4: write-through on every
    // other XCD's workgroups (nt 3, see the kernel) -- plain / write-through / mixed 23.3 / 23.7 / 22.7 us at order 2, 26.1 / 25.6 / 25.5 at 4,
    // 26.2 / 27.2 / 26.5 at 6 (plain stays)
    if (ctx->opt_nt < 0 && p.q < 4 && !ctx->res_capture) p.nt = (p.nt == 2 && p.q <= 2) ? 3 : 0;
    // (the resident evaluator: write-through everywhere -- every request ends with a write-back of the L2, and dirty lines of plain stores make
    //  that 10 us per request: 41 against 32 us per evaluation at config 3)
    if (ctx->opt_nt < 0 && ctx->res_capture) p.nt = 2;
    const long long units = p.contig ? bk * d : bk * p.S;
    if (units > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
    const long long g = ctx->opt_grid > 0 ? std::min<long long>(ctx->opt_grid, units) : std::min<long long>(units, ncu);
    const size_t lds = v4_lds_bytes(d, m, np) + (ticket ? 8 * 8 * 128 : 0);  // (+ the dispatcher's ring of controls and steps)
    // tiles of the ring in use: q - 1 when workgroups walk several items (the P wave must not run a whole item ahead: measured
    // 8 % on 8 trajectories per launch), all the module has for one-item launches (0.4 us there); option v4_power_tiles overrides
    p.v4_np = ctx->opt_v4_np > 0 ? (int)std::min<int64_t>(ctx->opt_v4_np, np) : ticket ? np : (units > g ? std::max(1, std::min(np, p.q - 1)) : std::min(np, p.q));
    // the cooperative first item (waves 0-3 build the powers, the stream waves fold) with a ring shorter than q: pays up to order 8 (one
    // trajectory 31.4 -> 29.3-30.5 us), not at order 10 (five powers through three tiles: 35.5 against 34.1 us)
    // (since round 5 the first item's powers beyond the ring borrow the chains' dW tiles -- q - v4_np <= m -- and every order starts cooperatively)
    if (p.q >= 5 && p.v4_np < p.q && (p.q - p.v4_np > m || (p.v4_flags & 16))) p.v4_flags |= 4;
    if (ticket) p.tail_mode = p.tail_mode == 3 ? 0 : p.tail_mode;  // (the writer wave stores delta and the tails of a chain ticket)
    if (want_merit && (p.tail_mode == 3 || ticket)) {
        if (!ctx->dmcols) HIP_TRY(ctx, hipMalloc((void **)&ctx->dmcols, (size_t)ctx->desc.batch * p.K * p.d * (p.m + 2) * sizeof(double)));
        p.mpart = ctx->dmcols;
        p.mlam = ctx->merit_lam;
        ctx->merit_fused = 1;
    }
    const double *tab = ctx->dv4_tab + (ctx->desc.per_member_G0 ? (long long)ctx->win_first * v4.n_drift_pad : 0);
    const double *dcf = ctx->dv4_dcf + (ctx->desc.per_member_G0 ? (long long)ctx->win_first * v4.n_dcf_pad : 0);
    void *args[] = {(void *)&p, (void *)&tab, (void *)&ctx->dv4_mags, (void *)&dcf};
    hipFunction_t fk = ctx->v4_f;
    if (ticket) {  // (a module of its own: the static launches -- one trajectory: the start-up counts -- run without the ticket roles' code)
        if (!ctx->v4_ft && !ctx->v4_ft_failed) {
            const std::string src = v4_source(*ctx->v4_plan, p.q, np, (int)ctx->opt_v4_variant, 1);
            const std::string key = "fused-sparse-tickets:" + std::to_string(p.q) + ":" + std::to_string(std::hash<std::string>{}(src));
            ctx->v4_ft = jit_compile(ctx->device, key, src, "pcl_fused_sparse_kernel", true);
            if (!ctx->v4_ft) {
                ctx->v4_ft_failed = 1;
                // the static-split module computes the same bits: noted once, an error only under require_jit
                if (jit_fell_back(ctx, "residual + Jacobian (slice tickets; the static split runs instead)") == PCL_EHIP) return PCL_EHIP;
            }
        }
        if (ctx->v4_ft)
            fk = ctx->v4_ft;
        else
            return launch_fused_v4_static(ctx, p, compact, want_merit);
    }
#ifdef PCL_LAB
    if (ctx->res_capture) {  // pcl_resident_start: the launch as data
        ctx->res.p = p, ctx->res.tab = tab, ctx->res.dcf = dcf, ctx->res.grid = g, ctx->res.lds = lds, ctx->res.block = 64u * (unsigned)(m + 9);
        ctx->res_capture = false;
        return PCL_OK;
    }
#endif
    bool timed = false;
    if (tune && tune_slot >= 0) {  // one timed sample of the per-array choice
        bool ok = true;
        for (int e = 0; e < 2 && ok; ++e)
            if (!tune->ev[tune_slot][e]) ok = hipEventCreate(&tune->ev[tune_slot][e]) == hipSuccess;
        timed = ok && hipEventRecord(tune->ev[tune_slot][0], ctx->stream) == hipSuccess;
    }
    HIP_TRY(ctx, hipModuleLaunchKernel(fk, (unsigned)g, 1, 1, 64 * (m + 9 + (ticket ? 1 : 0)), 1, 1, (unsigned)lds, ctx->stream, args, nullptr));
    if (timed && hipEventRecord(tune->ev[tune_slot][1], ctx->stream) == hipSuccess) {
        tune->pend[tune_slot] = true;
        tune->pend_variant[tune_slot] = ticket ? 1 : 0;
    }
    ctx->last_kernel = 40 + p.q;
    ctx->last_n_stream = 0;
    if (ticket) ctx->ticket_launched = true;
    return PCL_OK;
}

static int launch_pade_general(pcl_ctx *ctx, KParams &p, bool want_jac) {
    fill_pade(p, ctx->desc.pade_order);
    if (want_jac && ctx->opt_general_version != 1) {
        const int rc = launch_pade_v2(ctx, p);
        if (rc != PCL_ENOTIMPL) return rc;
        if (ctx->opt_general_version == 2) return fail(ctx, PCL_ESHAPE, "the lock-step general-order kernel does not take this shape (n = %d, %d columns, m = %d)", p.n, p.cols, p.m);
    }
    p.nc = ctx->opt_cols_per_slice > 0 ? (int)std::min<int64_t>(ctx->opt_cols_per_slice, p.cols) : p.cols;
    while (p.nc > 1 && pade_lds_bytes(p, want_jac) > (size_t)ctx->max_lds) --p.nc;
    const size_t lds = pade_lds_bytes(p, want_jac);
    if (lds > (size_t)ctx->max_lds)
        return fail(ctx, PCL_ESHAPE, "general-order kernel needs %zu B of LDS (> %d) for d=%d, m=%d, order %d", lds, ctx->max_lds, p.d, p.m, 2 * p.q);
    p.S = (p.cols + p.nc - 1) / p.nc;
    const long long grid = (long long)p.batch * p.K * p.S;
    if (grid > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
    typedef void (*kern_t)(const KParams);
    kern_t kern = want_jac ? (kern_t)pcl_pade_kernel<true> : (kern_t)pcl_pade_kernel<false>;
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)ctx->opt_general_threads), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_kernel = 90 + p.q;
    ctx->last_n_stream = 0;
    return PCL_OK;
}

static int resolve_order(pcl_ctx *ctx, const double *Z_host, const char *where);
static int launch_fused(pcl_ctx *ctx, const double *Z, double *delta, double *jac, bool compact) {
    ON_DEVICE(ctx);
    if (int rc = check_device_error(ctx, "pcl_eval / pcl_jac")) return rc;
    if (int rc = resolve_order(ctx, nullptr, "pcl_eval / pcl_jac")) return rc;
    KParams p;
    fill_params(ctx, p);
    p.Z = Z + (ctx->desc.batch_mode == PCL_BATCH_TRAJ ? (long long)ctx->win_first * ctx->desc.z_dim * ctx->desc.N : 0);
    p.delta = delta;
    p.jac = jac;
    p.compact = compact ? 1 : 0;
    p.jac_per = compact ? jac_per_compact(ctx) : jac_per_full(ctx);
    const bool want_jac = jac != nullptr;
    // streaming stores of the Jacobian blocks (auto): write-through while the launch's values fit the infinity cache with room to
    // spare (one trajectory of config 3: 133 MB), plain write-back above (see store2)
    if (ctx->opt_nt < 0 && want_jac && !compact && (long long)ctx->win_count * ctx->K * jac_per_full(ctx) * 8 <= (192LL << 20)) p.nt = 2;
    // kernel_version 4: the pattern-compiled fused kernel (sparse iso generators, any order).  auto: every order -- at order 4 it measures
    // 27.2 against 27.4 us for one trajectory, 184 against 190 us for 8 and 7.8 against 10.6 us per evaluation for the compact launches
    // of the host-delivery path (kernel 3 on the same box), and one kernel family behind every entry point keeps the full and the compact
    // values bitwise equal.  Kernel 3 keeps the payload-fused call and contexts that set its own switches.
    const bool v4_auto = ctx->opt_kernel == 0 && (ctx->desc.pade_order != 4 || (ctx->opt_stream_wg < 0 && ctx->opt_use_mfma != 0)) && !ctx->opt_general &&
                         ctx->opt_general_version == 0;
    if (want_jac && (ctx->opt_kernel == 4 || v4_auto)) {
        // the payload-fused call (pcl_eval_jac_merit_dev): kernel 4's otherwise idle writer wave forms the payload's dot products per
        // state column while the item's vectors are in their tiles -- any order
        const bool want_merit = ctx->merit_want && !compact && delta && ctx->win_first == 0 && ctx->win_count == ctx->desc.batch;
        const int rc = launch_fused_v4(ctx, p, compact, want_merit);
        if (rc != PCL_ENOTIMPL) return rc;
        if (ctx->opt_kernel == 4) return fail(ctx, PCL_ESHAPE, "kernel_version=4 needs sparse exact-iso generators of a unitary problem (9 <= d, tiles within LDS), 1..6 drives and jit=1 (%s)", g_jit_note.c_str());
    }
#ifdef PCL_LAB
    if (ctx->res_capture) return fail(ctx, PCL_ESHAPE, "pcl_resident_start: the resident evaluator is kernel 4's (sparse exact-iso generators of a unitary problem, 9 <= d, tiles within LDS, 1..6 drives, jit = 1; kernel_version 0 or 4)");
#endif
    if (p.nt == 3) p.nt = 2;  // (the mixed mode is kernel 4's)
    // residual only on the same products (eval_kernel 3; auto: every order -- measured against the other residual kernels)
    if (!want_jac && (ctx->opt_eval_kernel == 3 || (ctx->opt_eval_kernel == 0 && ctx->opt_kernel == 0 && !ctx->opt_general && ctx->opt_general_version == 0))) {
        const int rc = launch_eval_v4(ctx, p);
        if (rc != PCL_ENOTIMPL) return rc;
        if (ctx->opt_eval_kernel == 3) return fail(ctx, PCL_ESHAPE, "eval_kernel=3 needs sparse exact-iso generators of a unitary problem (9 <= d, tiles within LDS), 1..6 drives and jit=1 (%s)", g_jit_note.c_str());
    }
    // kernel_version 5 (auto wherever it applies: n <= 16 rows, at most 8 state columns, m <= 8 -- BASELINE configs 1 and 2, every system of the
    // reference's docs with d <= 8; every order): one wave per interval, one round of global loads, everything else in registers and a few KB
    // of LDS (pcl_kernel_fused_small.hpp)
    // (9 .. 16 rows: four matrix entries per lane -- residual only 5.3 against 5.9 us at d = 5, but residual + Jacobian 11.2 against the 8.3 us of
    //  kernel 1, so `auto` takes the 16-row instance for the residual alone; kernel_version = 5 forces it)
    if ((ctx->opt_kernel == 5 || (ctx->opt_kernel == 0 && !ctx->opt_general && ctx->opt_general_version == 0 && (ctx->n <= 8 || !want_jac))) && ctx->n <= 16 &&
        ctx->cols <= 8 && p.m <= 8 && (p.m == 0 || ctx->dGjd)) {  // (the payload-fused call: this kernel + the separate payload kernels)
        fill_pade(p, ctx->desc.pade_order);
        const long long items = (long long)p.batch * p.K;
        if (items > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
        const long long grid = ctx->opt_grid > 0 ? std::min<long long>(ctx->opt_grid, items) : std::min<long long>(items, 16LL * std::max(ctx->n_cu, 1));
        typedef void (*ksm_t)(const KParams, const double *);
        const ksm_t ksm = ctx->n <= 8 ? (want_jac ? (ksm_t)pcl_fused_small_kernel<true, 8> : (ksm_t)pcl_fused_small_kernel<false, 8>)
                                      : (want_jac ? (ksm_t)pcl_fused_small_kernel<true, 16> : (ksm_t)pcl_fused_small_kernel<false, 16>);
        hipLaunchKernelGGL(ksm, dim3((unsigned)grid), dim3(64), 0, ctx->stream, p, (const double *)ctx->dGjd);
        HIP_TRY(ctx, hipGetLastError());
        ctx->last_kernel = 50 + p.q;
        ctx->last_n_stream = 0;
        return PCL_OK;
    }
    if (ctx->opt_kernel == 5) return fail(ctx, PCL_ESHAPE, "kernel_version=5 (the small-system kernel) needs n <= 16 rows, at most 8 state columns and at most 8 drives (n = %d, cols = %d, m = %d)", ctx->n, ctx->cols, p.m);
    if (ctx->desc.pade_order != 4 || ctx->opt_general || ctx->vec) return launch_pade_general(ctx, p, want_jac);
    p.ell_lds = ell_fits_lds(ctx) ? 1 : 0;
    // auto: kernel 3 where its shape-specialised instance applies (BASELINE configs 3/4/5); its run-time-shape instances
    // lose to kernels 1 / 2 on every other shape measured (scripts/small_d_probe*.py: up to 3x), so they need kernel_version = 3
    if (want_jac && (!compact || v3_role_split_fits(ctx)) && (ctx->opt_kernel == 3 || (ctx->opt_kernel == 0 && v3_specialised(ctx))) &&
        ctx->opt_use_mfma != 0 && v3_supported(ctx) && !ctx->vec && ctx->cols == ctx->desc.d) {
        // default: contiguous column ranges (one item per interval touched); an explicit cols_per_slice or
        // contiguous = 0 selects the round-robin slices
        p.contig = (compact || v3_contiguous(ctx)) ? 1 : 0;  // compact: contiguous ranges, every workgroup in the matrix role
        p.all_matrix = compact ? 1 : 0;
        p.nc = p.contig ? p.d : choose_cols_v3(ctx);
        if (!p.contig && ctx->opt_cols_per_slice <= 0 && p.nc < 2 && p.d >= 2) p.nc = 2;  // keep the 2-column chunks of the specialised instance
        p.ncw = v3_ncw(ctx, p.nc);
        p.tab_lds = 1;
        size_t lds3 = fused3_lds_bytes(ctx, p, true);
        if (lds3 > (size_t)ctx->max_lds) {  // large union / ELL tables stay in memory
            p.tab_lds = 0;
            lds3 = fused3_lds_bytes(ctx, p, false);
        }
        if (lds3 > (size_t)ctx->max_lds) goto not_v3;  // double-buffered tiles do not fit (n close to 64): kernel v2
        p.S = (p.d + p.nc - 1) / p.nc;
        const long long items = (long long)p.batch * p.K * p.S;
        if (items > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
        typedef void (*kern3_t)(const KParams);
        const int ewr = (p.m <= 8 && ctx->ell_w >= 1 && ctx->ell_w <= 2) ? ctx->ell_w : 0;
        kern3_t kern3 = ewr == 1 ? (kern3_t)pcl_fused_kernel_v3<1, 0, 0, 0> : ewr == 2 ? (kern3_t)pcl_fused_kernel_v3<2, 0, 0, 0> : (kern3_t)pcl_fused_kernel_v3<0, 0, 0, 0>;
        // shape-specialised instances (compile-time d, m, chunk width; two drive entries per row):
        //   three 3-level transmons (BASELINE configs 3/4/5), two 5-level transmons, two 4-level transmons
        bool spec3 = false;
        // pcl_eval_jac_merit_dev: the MERIT instances (built in for config 3's shape, compiled on first use for other shapes)
        bool want_merit = ctx->merit_want && !compact && delta && ctx->win_first == 0 && ctx->win_count == ctx->desc.batch;
        if (ewr == 2 && p.ncw == 2 && ctx->opt_specialize) {
            spec3 = true;
            if (p.d == 27 && p.m == 6) kern3 = want_merit ? (kern3_t)pcl_fused_kernel_v3<2, 27, 6, 2, true> : (kern3_t)pcl_fused_kernel_v3<2, 27, 6, 2>;
            else if (p.d == 25 && p.m == 4 && !want_merit) kern3 = (kern3_t)pcl_fused_kernel_v3<2, 25, 4, 2>;  // (merit: compiled on first use)
            else if (p.d == 16 && p.m == 4 && !want_merit) kern3 = (kern3_t)pcl_fused_kernel_v3<2, 16, 4, 2>;
            else spec3 = false;
        }
        hipFunction_t jitf = nullptr;  // run-time compiled instance for this context's shape
        if (!spec3 && ctx->opt_jit && ctx->opt_specialize && ewr >= 1) {
            char inst[96];
            snprintf(inst, sizeof inst, want_merit ? "pcl_fused_kernel_v3<%d, %d, %d, %d, true>" : "pcl_fused_kernel_v3<%d, %d, %d, %d>", ewr, p.d, p.m, p.ncw);
            jitf = jit_function(ctx->device, inst);
        }
        if (!spec3 && !jitf && ctx->opt_kernel == 0) goto not_v3;  // auto never runs the run-time-shape instances
        ctx->last_kernel = jitf ? 32 : 30 + (spec3 ? 1 : 0);
        if (!jitf) {
            int rc = set_lds_attr(ctx, (const void *)kern3, (want_merit && spec3) ? 8 : 6, lds3);
            if (rc != PCL_OK) return rc;
        }
        const long long units = p.contig ? (long long)p.batch * p.K * p.d : items;  // what the grid is cut into
        const long long g3 = ctx->opt_grid > 0 ? std::min<long long>(ctx->opt_grid, units) : std::min<long long>(units, std::max(ctx->n_cu, 1));
        p.n_stream = 0;
        if (p.contig && !p.all_matrix && g3 >= 2 && v3_role_split_fits(ctx)) {
            const long long want = ctx->opt_stream_wg < 0 ? g3 / 2 : ctx->opt_stream_wg;  // auto: half the workgroups stream
            if (want > 0) p.n_stream = (int)std::min<long long>(want, g3 - 1);
        }
        ctx->last_n_stream = p.n_stream;
        if (want_merit && (spec3 || jitf)) {
            // pcl_eval_jac_merit_dev: the matrix waves also leave the reduce payload's dot products per state column
            if (!ctx->dmcols) HIP_TRY(ctx, hipMalloc((void **)&ctx->dmcols, (size_t)ctx->desc.batch * p.K * p.d * (p.m + 2) * sizeof(double)));
            p.mpart = ctx->dmcols;
            p.mlam = ctx->merit_lam;
            ctx->merit_fused = 1;
        }
        if (jitf) {
            void *args[] = {(void *)&p};
            HIP_TRY(ctx, hipModuleLaunchKernel(jitf, (unsigned)g3, 1, 1, 512, 1, 1, (unsigned)lds3, ctx->stream, args, nullptr));
            return PCL_OK;
        }
        hipLaunchKernelGGL(kern3, dim3((unsigned)g3), dim3(512), lds3, ctx->stream, p);
        HIP_TRY(ctx, hipGetLastError());
        return PCL_OK;
    }
not_v3:
    // residual only (what the solver calls in every line-search trial), sparse iso generators: the pattern-compiled kernel
    // (auto: launches with more intervals than CUs -- below that one wave per interval is a longer chain than the matrix-core kernel's)
    if (!want_jac && (ctx->opt_eval_kernel == 2 || (ctx->opt_eval_kernel == 0 && (long long)p.batch * p.K > ctx->n_cu)) && ctx->sp_plan && !ctx->sp_failed && ctx->opt_jit &&
        (ctx->opt_kernel == 0 || ctx->opt_kernel >= 3)) {
        const pcl_codegen::SpPlan &sp = *ctx->sp_plan;
        const long long items = (long long)p.batch * p.K;
        if (items > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
        if (!ctx->sp_feval) {
            const std::string src = sparse_source(sp);
            const std::string key = "sparse:" + std::to_string(std::hash<std::string>{}(src));
            ctx->sp_feval = jit_compile(ctx->device, key, src, "pcl_eval_sparse_kernel", true);
            if (!ctx->sp_feval) {
                ctx->sp_failed = 1;
                if (int rc = jit_fell_back(ctx, "residual"); rc != PCL_ENOTIMPL) return rc;
            }
        }
        if (ctx->sp_feval) {
            if (ctx->sp_gvals_cap < (long long)ctx->desc.batch * p.K) {
                if (ctx->dsp_gvals) (void)hipFree(ctx->dsp_gvals);
                ctx->dsp_gvals = nullptr;
                ctx->sp_gvals_cap = 0;
                HIP_TRY(ctx, hipMalloc((void **)&ctx->dsp_gvals, ((size_t)ctx->desc.batch * p.K * sp.nzp + 32) * sizeof(double)));
                ctx->sp_gvals_cap = (long long)ctx->desc.batch * p.K;
            }
            // one wave per interval; the intervals are spread over the CUs first, then over the waves of a workgroup (<= 8)
            const long long ncu = std::max(ctx->n_cu, 1);
            int nw = (int)std::min<long long>(8, std::max<long long>(1, (items + ncu - 1) / ncu));
            long long grid = std::min<long long>((items + nw - 1) / nw, ncu);
            if (ctx->opt_grid > 0) grid = std::min<long long>(ctx->opt_grid, (items + nw - 1) / nw);
            const size_t ldse = (size_t)nw * (sp.n + 1) * sp.d * sizeof(double);
            void *args[] = {(void *)&p, (void *)&ctx->dsp_gvals, (void *)&ctx->dsp_pos_n, (void *)&ctx->dsp_coef_n};
            HIP_TRY(ctx, hipModuleLaunchKernel(ctx->sp_feval, (unsigned)grid, 1, 1, 64 * nw, 1, 1, (unsigned)ldse, ctx->stream, args, nullptr));
            ctx->last_kernel = 70;  // the pattern-compiled residual kernel
            ctx->last_n_stream = 0;
            return PCL_OK;
        }
        if (ctx->opt_eval_kernel == 2) return fail(ctx, PCL_ESHAPE, "eval_kernel=2: the pattern-compiled kernel is not available (%s)", g_jit_note.c_str());
    } else if (!want_jac && ctx->opt_eval_kernel == 2) {
        return fail(ctx, PCL_ESHAPE, "eval_kernel=2 needs sparse iso generators, a unitary problem with 9 <= d <= 32, 1..6 drives and jit=1");
    }
    // residual only, any generators: the dedicated matrix-core kernel for unitary states
    if (!want_jac && ctx->opt_use_mfma != 0 && !ctx->vec && ctx->cols == ctx->desc.d && ctx->desc.d >= 9 && (ctx->opt_kernel == 0 || ctx->opt_kernel >= 3)) {
        typedef void (*kerne_t)(const KParams);
        const int wu = (ctx->uell_w <= 2 && ctx->n_upos <= 256 * PCL_NUE_EV) ? ctx->uell_w : -1;
        const bool spec = ctx->opt_specialize && p.d == 27;
        kerne_t ke = wu == 1 ? (spec ? (kerne_t)pcl_eval_kernel<1, 27> : (kerne_t)pcl_eval_kernel<1, 0>)
                   : wu == 2 ? (spec ? (kerne_t)pcl_eval_kernel<2, 27> : (kerne_t)pcl_eval_kernel<2, 0>)
                             : (kerne_t)pcl_eval_kernel<-1, 0>;
        const size_t lde = (p.d & 1) ? (size_t)p.n : (size_t)p.LD;  // the kernel's leading dimension (2*odd)
        const size_t ldse = (lde * p.n + 4 * lde * 16 + 2 * (size_t)(p.m + 1) + 2) * sizeof(double) + 128;
        if (ldse <= (size_t)ctx->max_lds) {
            if (int rc = set_lds_attr(ctx, (const void *)ke, 5, ldse)) return rc;
            const long long items = (long long)p.batch * p.K;
            if (items > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
            // two resident workgroups per CU measured best (three: 28 us per 8 trajectories, two: 21.5); every workgroup walks the
            // same number of intervals: grid = items / rounds
            const int per_cu = std::max(1, std::min(2, (int)((size_t)ctx->max_lds / ldse)));
            const long long slots = (long long)per_cu * std::max(ctx->n_cu, 1), rounds = (items + slots - 1) / slots;
            const long long ge = ctx->opt_grid > 0 ? std::min<long long>(ctx->opt_grid, items) : (items + rounds - 1) / rounds;
            ctx->last_kernel = 60 + ((spec && wu > 0) ? 1 : 0);
            ctx->last_n_stream = 0;
            hipLaunchKernelGGL(ke, dim3((unsigned)ge), dim3(256), ldse, ctx->stream, p);
            HIP_TRY(ctx, hipGetLastError());
            return PCL_OK;
        }
    }
    // auto: small Hilbert dimensions are launch- / latency-bound, one workgroup per item (kernel 1) beats the persistent
    // kernels there (measured: d <= 8 always, d <= 16 while all items fit one round of workgroups)
    const bool v1_auto = ctx->opt_kernel == 0 && (ctx->desc.d <= 8 || (ctx->desc.d <= 16 && (long long)ctx->win_count * ctx->K <= 512));
    const bool v2 = !v1_auto && (ctx->opt_kernel == 0 || ctx->opt_kernel >= 2) && ctx->opt_use_mfma != 0;
    const bool unitary = !ctx->vec && ctx->cols == ctx->desc.d;  // kernel 3 and the specialised instances assume X is n x d
    p.nc = choose_cols_per_slice(ctx, want_jac);
    auto bytes = [&]() { return v2 ? fused2_lds_bytes(p, want_jac, p.ell_lds != 0) : fused_lds_bytes(p, want_jac); };
    size_t lds = bytes();
    while (lds > (size_t)ctx->max_lds && p.nc > 1) {
        p.nc = (p.nc + 1) / 2;
        lds = bytes();
    }
    if (lds > (size_t)ctx->max_lds) return fail(ctx, PCL_ESHAPE, "fused kernel needs %zu B of LDS (> %d)", lds, ctx->max_lds);
    p.S = (p.cols + p.nc - 1) / p.nc;
    const long long grid = (long long)p.batch * p.K * p.S;
    if (grid > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "grid too large");
    if (v2) {
        typedef void (*kern_t)(const KParams);
        const int wu = (ctx->uell_w <= 2 && ctx->n_upos <= 1024 && !ctx->desc.per_member_G0) ? ctx->uell_w : -1;
        kern_t kern = want_jac ? (wu == 1 ? (kern_t)pcl_fused_kernel_v2<true, 1, 0, 0, 0> : wu == 2 ? (kern_t)pcl_fused_kernel_v2<true, 2, 0, 0, 0> : (kern_t)pcl_fused_kernel_v2<true, -1, 0, 0, 0>)
                               : (wu == 1 ? (kern_t)pcl_fused_kernel_v2<false, 1, 0, 0, 0> : wu == 2 ? (kern_t)pcl_fused_kernel_v2<false, 2, 0, 0, 0> : (kern_t)pcl_fused_kernel_v2<false, -1, 0, 0, 0>);
        // shape-specialised instance (BASELINE config 3/4/5: three 3-level transmons, d = 27, six drives, 3-column slices)
        if (want_jac && unitary && wu == 1 && p.d == 27 && p.m == 6 && p.nc == 3 && ctx->opt_specialize) kern = (kern_t)pcl_fused_kernel_v2<true, 1, 27, 6, 3>;
        ctx->last_n_stream = 0;
        ctx->last_kernel = 20 + ((unitary && wu == 1 && p.d == 27 && p.m == 6 && p.nc == 3 && ctx->opt_specialize && want_jac) ? 1 : 0);
        {
            int rc = set_lds_attr(ctx, (const void *)kern, want_jac ? 4 : 5, lds);  // wu is fixed per context
            if (rc != PCL_OK) return rc;
        }
        // persistent grid: as many workgroups as are resident at once
        const int per_cu = std::max(1, std::min(2, (int)((size_t)ctx->max_lds / lds)));
        const long long resident = (long long)per_cu * std::max(ctx->n_cu, 1);
        const long long g2 = ctx->opt_grid > 0 ? std::min<long long>(ctx->opt_grid, grid) : std::min(grid, resident);
        hipLaunchKernelGGL(kern, dim3((unsigned)g2), dim3(512), lds, ctx->stream, p);
    } else {
        const bool mf = ctx->opt_use_mfma != 0;
        auto kern = want_jac ? (mf ? pcl_fused_kernel<true, true> : pcl_fused_kernel<true, false>)
                             : (mf ? pcl_fused_kernel<false, true> : pcl_fused_kernel<false, false>);
        int rc = set_lds_attr(ctx, (const void *)kern, (want_jac ? 0 : 2) + (mf ? 0 : 1), lds);
        if (rc != PCL_OK) return rc;
        ctx->last_kernel = 10;
        ctx->last_n_stream = 0;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx->stream, p);
    }
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}

static size_t hess_lds_bytes(const KParams &p) {
    const size_t nscal = (size_t)(p.m + 1) * (p.m + 2) / 2;
    return ((size_t)p.LD * p.n + (6 + 3 * (size_t)p.m) * p.LD * p.nc + 8 + p.m + 5 * nscal) * sizeof(double);
}

template <int EW, bool ANTI>
static const void *hess_v2_kernel(int m) {
    switch (m) {
    case 1: return (const void *)pcl_hess_kernel_v2<EW, 1, 0, ANTI>;
    case 2: return (const void *)pcl_hess_kernel_v2<EW, 2, 0, ANTI>;
    case 3: return (const void *)pcl_hess_kernel_v2<EW, 3, 0, ANTI>;
    case 4: return (const void *)pcl_hess_kernel_v2<EW, 4, 0, ANTI>;
    case 5: return (const void *)pcl_hess_kernel_v2<EW, 5, 0, ANTI>;
    case 6: return (const void *)pcl_hess_kernel_v2<EW, 6, 0, ANTI>;
    }
    return nullptr;
}
#define PCL_HESS_EW 2
static bool hess_v2_supported(const pcl_ctx *ctx) {
    const int m = ctx->desc.n_drives;
    return !ctx->vec && m >= 1 && m <= 6 && ctx->ell_w <= PCL_HESS_EW && ctx->ellt_w <= PCL_HESS_EW;
}
static size_t hess2_lds_bytes(const KParams &p) {
    const size_t nscal = (size_t)(p.m + 1) * (p.m + 2) / 2;
    return ((size_t)p.LD * p.n + 7 * (size_t)p.LD * 16 + 4 * nscal + 4 + 2) * sizeof(double);
}

static int launch_hess(pcl_ctx *ctx, const double *Z, const double *mu, double *hess) {
    ON_DEVICE(ctx);
    if (int rc = check_device_error(ctx, "pcl_hess")) return rc;
    if (int rc = resolve_order(ctx, nullptr, "pcl_hess")) return rc;
    KParams p;
    fill_params(ctx, p);
    p.Z = Z + (ctx->desc.batch_mode == PCL_BATCH_TRAJ ? (long long)ctx->win_first * ctx->desc.z_dim * ctx->desc.N : 0);
    p.mu = mu;
    p.hess = hess;
    const bool mf = ctx->opt_use_mfma != 0;
    // hess_kernel 8 (auto at every order but 4): the pattern-compiled kernel with ONE WAVE PER GROUP OF STATE COLUMNS (pcl_kernel_hess_cols.hpp): every
    // wave a workgroup of its own with all m + 1 chains of 32 / (m + 1) columns, the scalar entries assembled by the wave of the interval
    // that arrives last
    // (order 4: the tuned kernel 6 below stays ahead from two trajectories per launch on -- 22.6 against 23.8 us, 56 against 60 at eight; one
    //  trajectory, i.e. at most n_cu / 2 intervals: 18.4 against 20.0 us -- kernel 6 needs its value-table launch first)
    const bool cols_auto = ctx->opt_hess_kernel == 0 && !ctx->opt_general &&
                           (ctx->desc.pade_order != 4 || 2LL * p.batch * p.K <= std::max(ctx->n_cu, 1));
    if ((ctx->opt_hess_kernel == 8 || cols_auto) && v4_available(ctx) && !ctx->v4_hessc_failed && p.m >= 1) {
        const pcl_codegen::V4Plan &v4 = *ctx->v4_plan;
        fill_pade(p, ctx->desc.pade_order);
        const long long items = (long long)p.batch * p.K;
        const int cpw = 32 / (p.m + 1), ng = (p.d + cpw - 1) / cpw;
        if (items * ng > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
        const size_t ldsc = hess_cols_lds_bytes(p.d, p.m, p.q, pcl_codegen::v4_gather_total(v4));
        if (!ctx->v4_fhessc && ldsc <= (size_t)ctx->max_lds) {
            const std::string src = v4_hess_cols_source(v4, p.q, (int)ctx->opt_v4_variant);
            const std::string key = "hess-cols:" + std::to_string(p.q) + ":" + std::to_string(std::hash<std::string>{}(src));
            ctx->v4_fhessc = jit_compile(ctx->device, key, src, "pcl_hess_cols_kernel", true);
            if (ctx->v4_fhessc) ctx->v4_fhessp = jit_compile(ctx->device, key, src, "pcl_hess_cols_pair_kernel", true);
            if (!ctx->v4_fhessc) {
                ctx->v4_hessc_failed = 1;
                if (int rc = jit_fell_back(ctx, "Hessian of the Lagrangian (column groups)"); rc != PCL_ENOTIMPL) return rc;
            }
        }
        if (ctx->v4_fhessc) {
            const long long cap = (long long)ctx->desc.batch * ctx->K;
            if (ctx->hc_cap < cap) {
                if (ctx->dhcx) (void)hipFree(ctx->dhcx);
                if (ctx->dhcc) (void)hipFree(ctx->dhcc);
                ctx->dhcx = nullptr, ctx->dhcc = nullptr, ctx->hc_cap = 0;
                HIP_TRY(ctx, hipMalloc((void **)&ctx->dhcx, (size_t)cap * ng * (p.m + 1) * (p.m + 1) * sizeof(double)));
                HIP_TRY(ctx, hipMalloc((void **)&ctx->dhcc, (size_t)cap * sizeof(unsigned int)));
                HIP_TRY(ctx, hipMemsetAsync(ctx->dhcc, 0, (size_t)cap * sizeof(unsigned int), ctx->stream));
                ctx->hc_cap = cap;
            }
            const long long wo = ctx->desc.per_member_G0 ? (long long)ctx->win_first : 0;
            const double *tab = ctx->dv4_tab + wo * v4.n_drift_pad, *tab_t = ctx->dv4_tab_t + wo * v4.n_drift_pad, *dcf = ctx->dv4_dcf + wo * v4.n_dcf_pad;
            ctx->ticket_launched = true;  // (arrival counters + exchange rows: a launch on another stream must not overlap this one)
            // Launches of several trajectories: the chain R_{q-2} .. R_1 is formed ONCE per interval by R-chain waves -- the first `intervals`
            // workgroups of the same launch (pcl_kernel_hess_cols.hpp: hc_rchain_role), lane = (half, column) -- instead of inside each of the interval's
            // column-group waves at 8 of 64 lanes: 15 % of an 8-seed launch at order 8 (profiles/r06_hess_ablations_4.log).  The tiles travel through
            // memory (written through; a counter per interval, reset by the interval's last column-group wave).  One trajectory keeps the chain inside
            // the waves (99 intervals: every wave starts at once, there is nothing to hide the chain waves behind).
            // option hess_rpre: -1 auto | 0 never | 1 R-chain waves  (the same waves as a launch of its own in front measured the same within 1 %: removed; profiles/r06_hess_rpre_*.log)
            // MEASURED (config 3, one box, alternating): 8 seeds 102-107 against 105 us at order 8, 130-134 against 136-138 at order 10; 64 seeds 630-640 against
            // 648-652 and 805-814 against 850-854 -- 3-5 %, not the 15 % the chain costs inside the waves: a chain wave is 20 k cycles of latency in a wave slot
            // and every column-group wave pays a flag and a tile round trip at its start.
            double *rpre = nullptr;
            unsigned int *rflag = nullptr;
            ctx->last_hess_rpre = 0;
            const size_t lds_r = (size_t)p.d * (p.n + 1) * sizeof(double);  // (one tile of d columns: the chain wave's exchange of the halves)
            // (auto: orders 8 and 10 -- two and three products in the chain; at order 6 the one product a wave saves is worth less than the chain waves cost:
            //  64 seeds 523 against 505 us; profiles/r06_hess_rpre_*.log)
            // (... and launches of more column-group waves than the device has wave slots -- 8 per CU: below that a launch is the latency of its waves and the
            //  chain waves only add to it: two trajectories at order 8 42.5 against 39.4 us)
            int rmode = ctx->opt_hess_rpre >= 0 ? (ctx->opt_hess_rpre ? 1 : 0) : ((items * ng > 8LL * std::max(ctx->n_cu, 1) && p.q >= 4) ? 1 : 0);
            if (p.q <= 2) rmode = 0;
            const int nx_ = ctx->opt_hess_xcd < 0 ? 8 : (int)std::max<int64_t>(1, ctx->opt_hess_xcd);
            // (with them a column-group wave holds ONE tile of the R_a -- the next pass's is copied in behind the current one's last use -- instead of q - 2:
            //  18.7 instead of 22.2 KB at config 3, order 10: eight waves per CU instead of seven)
            //  -- where all q - 2 fit without costing a wave per CU (order 8: 20.4 KB, eight either way) they are all copied in at the start: 2 % faster than tile by tile)
            const size_t lds1_ = std::max(hess_cols_lds_bytes(p.d, p.m, p.q, pcl_codegen::v4_gather_total(v4), 1), lds_r), ldsa_ = std::max(ldsc, lds_r);
            const int rt_all = std::min<size_t>(8, (size_t)ctx->max_lds / ldsa_) >= std::min<size_t>(8, (size_t)ctx->max_lds / lds1_) ? 1 : 0;
            const size_t ldsc1 = rt_all ? ldsa_ : lds1_;
            if (rmode && (ldsc1 > (size_t)ctx->max_lds || nx_ != 8)) rmode = 0;  // (the chain wave and its readers share an XCD's L2: blockIdx equal mod 8)
            if (rmode) {
                const size_t per_item = (size_t)(p.q - 2) * ng * hess_cols_rtile_doubles(p.d, p.m);  // [a][column group][tile as it lies in LDS]
                if (ctx->hcr_cap < cap * (long long)per_item || !ctx->dhcf) {
                    if (ctx->dhcr) (void)hipFree(ctx->dhcr);
                    if (ctx->dhcf) (void)hipFree(ctx->dhcf);
                    ctx->dhcr = nullptr, ctx->dhcf = nullptr, ctx->hcr_cap = 0;
                    HIP_TRY(ctx, hipMalloc((void **)&ctx->dhcr, (size_t)cap * per_item * sizeof(double)));
                    HIP_TRY(ctx, hipMemsetAsync(ctx->dhcr, 0, (size_t)cap * per_item * sizeof(double), ctx->stream));  // (a group past its last column: zeros, never written -- copied along)
                    HIP_TRY(ctx, hipMalloc((void **)&ctx->dhcf, (size_t)cap * sizeof(unsigned int)));
                    HIP_TRY(ctx, hipMemsetAsync(ctx->dhcf, 0, (size_t)cap * sizeof(unsigned int), ctx->stream));
                    ctx->hcr_cap = cap * (long long)per_item;
                }
                rpre = ctx->dhcr;
                rflag = ctx->dhcf;
                ctx->last_hess_rpre = 1;
            }
            const long long n_rblk = rflag ? (items + 7) / 8 * 8 : 0;  // (one chain wave per interval; a multiple of 8: the column-group waves keep their XCDs)
            p.n_stream = (int)n_rblk;
            void *args[] = {(void *)&p, (void *)&tab, (void *)&tab_t, (void *)&ctx->dv4_mags, (void *)&dcf, (void *)&ctx->dhcx, (void *)&ctx->dhcc, (void *)&rpre, (void *)&rflag, (void *)&rt_all};
            // the waves of an interval on ONE XCD (blockIdx equal mod 8: the lines their neighbouring output runs share merge in that XCD's L2);
            // option hess_xcd: -1 auto (on) | 0 blockIdx order | n: the modulus
            const int nx = ctx->opt_hess_xcd < 0 ? 8 : (int)std::max<int64_t>(1, ctx->opt_hess_xcd);
            p.S = nx;
            const long long grid_hc = n_rblk + (nx > 1 ? ((items + nx - 1) / nx) * nx * ng : items * ng);
            if (grid_hc > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
            // One trajectory per launch (at most n_cu / 2 intervals: every wave runs at once, the launch's time is ONE wave's latency): two waves per column
            // group -- the chain (product, gathers) in one, what a level contributes in the other, two buffers of chain slots between them.  Same values, bitwise.
            // (auto also asks that every workgroup of the launch is resident at once -- LDS decides: 47.5 KB at config 3, order 10: three per CU -- or the launch is two rounds of latencies)
            // MEASURED (config 3, one trajectory, one box, alternating; profiles/r06_hess_pair_*.log): order 4 17.6 -> 16.7 us, order 6 24.0 -> 20.4, order 8 29.1 -> 24.1,
            // order 10 36.0 -> 27.6.
            const size_t ldsp = ldsc + ((size_t)2 * (p.m + 1) * cpw * (p.n + 1) + 2) * sizeof(double);  // (HP_NBUF = 3 buffers of chain slots instead of one + the sync words)
            const bool pair = ctx->v4_fhessp && !rmode && ldsp <= (size_t)ctx->max_lds && (ctx->opt_hess_pair == 1 || (ctx->opt_hess_pair < 0 && p.q >= 2 && 2 * items <= std::max(ctx->n_cu, 1) && items * ng <= (long long)std::max(ctx->n_cu, 1) * (long long)((size_t)ctx->max_lds / ldsp)));  // (order 2: one pass with a product -- nothing to overlap: 13.4 against 13.2 us)
            ctx->last_hess_pair = pair ? 1 : 0;
            if (pair) {
                void *pargs[] = {(void *)&p, (void *)&tab, (void *)&tab_t, (void *)&ctx->dv4_mags, (void *)&dcf, (void *)&ctx->dhcx, (void *)&ctx->dhcc};
                HIP_TRY(ctx, hipModuleLaunchKernel(ctx->v4_fhessp, (unsigned)grid_hc, 1, 1, 128, 1, 1, (unsigned)ldsp, ctx->stream, pargs, nullptr));
            } else
            {
                static const size_t lds_pad = getenv("PCL_HC_LDS_PAD") ? (size_t)atol(getenv("PCL_HC_LDS_PAD")) : 0;  // (occupancy experiment: lab/probes/README.md, round 6)
                HIP_TRY(ctx, hipModuleLaunchKernel(ctx->v4_fhessc, (unsigned)grid_hc, 1, 1, 64, 1, 1, (unsigned)((rflag ? ldsc1 : ldsc) + lds_pad), ctx->stream, args, nullptr));
            }
            ctx->last_hess_kernel = 80 + p.q;
            ctx->last_hess_split = 0;
            return PCL_OK;
        }
        if (ctx->opt_hess_kernel == 8) return fail(ctx, PCL_ESHAPE, "hess_kernel=8: the column-group kernel is not available (%s)", g_jit_note.c_str());
    } else if (ctx->opt_hess_kernel == 8) {
        return fail(ctx, PCL_ESHAPE, "hess_kernel=8 needs sparse exact-iso generators of a unitary problem (9 <= d <= 32), 1..6 drives and jit=1");
    }
    // hess_kernel 7 (auto where kernel 8 is not available): the pattern-compiled kernel on the products of fused kernel 4 -- any order,
    // one workgroup of m + 1 waves per interval (column slices where the m + 3 + 2 (q - 2) tiles do not fit LDS)
    if ((ctx->opt_hess_kernel == 7 || (ctx->opt_hess_kernel == 0 && ctx->desc.pade_order != 4 && !ctx->opt_general)) && v4_available(ctx) && !ctx->v4_hess_failed) {
        const pcl_codegen::V4Plan &v4 = *ctx->v4_plan;
        fill_pade(p, ctx->desc.pade_order);
        // small launches (at most n_cu / 2 intervals: one trajectory): TWO workgroups per interval, each with half of the drive chains (and its
        // own copy of the W and power chains) -- one wave per SIMD, 512 registers per lane instead of 256 (no spilled values), the 28 scalar
        // entries assembled by the workgroup that arrives last (option hess_split: -1 auto | 0 | 1)
        const long long items = (long long)p.batch * p.K;
        if (items > 0x3fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
        const int nz = p.q > 2 ? p.q - 2 : 0;
        auto bytes_ = [&](int mh_, int nc) { return ((size_t)(mh_ + 3 + 2 * nz + (p.q == 2 ? 1 : 0) /* SH_NTILES */) * nc * (p.n + 1) + (size_t)(p.m + 1) * (p.m + 2) + 8) * sizeof(double); };
        auto cols_ = [&](int mh_) {
            int nc = p.d;
            while (nc > 1 && bytes_(mh_, nc) > (size_t)ctx->max_lds) nc = (nc + 1) / 2;
            return nc;
        };
        // ... and at every size where one workgroup's tiles need column slices and half the drive chains' do not (config 3 at order 10:
        // 15 tiles of 27 columns do not fit 160 KB, 12 do; 8 trajectories per launch 394 -> 267 us)
        const bool split = p.m >= 2 && (ctx->opt_hess_split == 1 || (ctx->opt_hess_split < 0 && (2 * items <= std::max(ctx->n_cu, 1) || (cols_(p.m) < p.d && cols_((p.m + 1) / 2) == p.d))));
        const int ns = split ? 2 : 1, mh = (p.m + ns - 1) / ns;
        auto bytes = [&](int nc) { return bytes_(mh, nc); };
        p.nc = cols_(mh);
        if (ctx->opt_cols_per_slice > 0) p.nc = (int)std::min<int64_t>(p.nc, ctx->opt_cols_per_slice);
        hipFunction_t &fh = split ? ctx->v4_fhess2 : ctx->v4_fhess;
        if (!fh) {
            const std::string src = v4_hess_source(v4, p.q, (int)ctx->opt_v4_variant, ns);
            const std::string key = "hess-sparse4:" + std::to_string(p.q) + ":" + std::to_string(ns) + ":" + std::to_string(std::hash<std::string>{}(src));
            fh = jit_compile(ctx->device, key, src, "pcl_hess_sparse4_kernel", true);
            if (!fh) {
                ctx->v4_hess_failed = 1;
                if (int rc = jit_fell_back(ctx, "Hessian of the Lagrangian"); rc != PCL_ENOTIMPL) return rc;
            }
        }
        if (fh && split && ctx->h4_cap < (long long)ctx->desc.batch * ctx->K) {
            if (ctx->dh4x) (void)hipFree(ctx->dh4x);
            if (ctx->dh4c) (void)hipFree(ctx->dh4c);
            ctx->dh4x = nullptr, ctx->dh4c = nullptr, ctx->h4_cap = 0;
            const long long cap = (long long)ctx->desc.batch * ctx->K;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->dh4x, (size_t)cap * 2 * (mh + 1) * (p.m + 2) * sizeof(double)));
            HIP_TRY(ctx, hipMalloc((void **)&ctx->dh4c, (size_t)cap * sizeof(unsigned int)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->dh4c, 0, (size_t)cap * sizeof(unsigned int), ctx->stream));
            ctx->h4_cap = cap;
        }
        if (fh && bytes(p.nc) <= (size_t)ctx->max_lds) {
            const long long units = items * ns;
            const long long slots = std::max(ctx->n_cu, 1), rounds = (units + slots - 1) / slots;
            long long grid = (units + rounds - 1) / rounds;  // every workgroup walks the same number of (interval, drive group) units
            if (ctx->opt_grid > 0) grid = std::min<long long>(units, ctx->opt_grid);
            const long long wo = ctx->desc.per_member_G0 ? (long long)ctx->win_first : 0;
            const double *tab = ctx->dv4_tab + wo * v4.n_drift_pad, *tab_t = ctx->dv4_tab_t + wo * v4.n_drift_pad, *dcf = ctx->dv4_dcf + wo * v4.n_dcf_pad;
            ctx->ticket_launched = true;
            void *args[] = {(void *)&p, (void *)&tab, (void *)&tab_t, (void *)&ctx->dv4_mags, (void *)&dcf, (void *)&ctx->dh4x, (void *)&ctx->dh4c};
            HIP_TRY(ctx, hipModuleLaunchKernel(fh, (unsigned)grid, 1, 1, 64 * (1 + mh), 1, 1, (unsigned)bytes(p.nc), ctx->stream, args, nullptr));
            ctx->last_hess_kernel = 70 + p.q;
            ctx->last_hess_split = split ? 1 : 0;
            return PCL_OK;
        }
        if (ctx->opt_hess_kernel == 7) return fail(ctx, PCL_ESHAPE, "hess_kernel=7: the pattern-compiled general-order kernel is not available (%s)", g_jit_note.c_str());
    } else if (ctx->opt_hess_kernel == 7) {
        return fail(ctx, PCL_ESHAPE, "hess_kernel=7 needs sparse exact-iso generators of a unitary problem (9 <= d <= 32), 1..6 drives and jit=1");
    }
    if (ctx->desc.pade_order != 4 || ctx->opt_general) {
        // any diagonal Pade order (and the cross-check of the order-4 kernels): the general-order kernel, correctness first
        p.q = ctx->desc.pade_order / 2;
        double f[16];
        f[0] = 1.0;
        for (int i = 1; i < 16; ++i) f[i] = f[i - 1] * i;
        for (int j = 0; j <= p.q; ++j) p.pc[j] = f[2 * p.q - j] * f[p.q] / (f[2 * p.q] * f[j] * f[p.q - j]);
        const size_t npair = (size_t)p.m * (p.m + 1) / 2, nsc = (size_t)(p.m + 1) * (p.m + 2) / 2;
        auto bytes = [&](int nc) { return ((size_t)p.LD * p.n + (6 + 4 * (size_t)p.m + 2 * npair) * p.LD * nc + 8 + p.m + 4 * nsc) * sizeof(double); };
        p.nc = p.cols;
        while (p.nc > 1 && bytes(p.nc) > (size_t)ctx->max_lds) p.nc = (p.nc + 1) / 2;
        const size_t ldsg = bytes(p.nc);
        if (ldsg > (size_t)ctx->max_lds)
            return fail(ctx, PCL_ESHAPE, "general-order Hessian kernel needs %zu B of LDS (> %d) for d=%d, m=%d", ldsg, ctx->max_lds, p.d, p.m);
        const long long gridg = (long long)p.batch * p.K;
        if (gridg > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
        auto kg = mf ? pcl_hess_general_kernel<true> : pcl_hess_general_kernel<false>;
        HIP_TRY(ctx, hipFuncSetAttribute((const void *)kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg));
        hipLaunchKernelGGL(kg, dim3((unsigned)gridg), dim3(256), ldsg, ctx->stream, p);
        HIP_TRY(ctx, hipGetLastError());
        ctx->last_hess_kernel = 90 + p.q;
        return PCL_OK;
    }
    // version 4 (default where it applies): the pattern-compiled kernel -- sparse iso generators, one state column per lane
    if ((ctx->opt_hess_kernel == 0 || ctx->opt_hess_kernel == 4) && ctx->sp_plan && !ctx->sp_failed && !ctx->sp_hess_unfit && ctx->opt_jit && ctx->desc.d == p.d) {
        const pcl_codegen::SpPlan &sp = *ctx->sp_plan;
        const long long items = (long long)p.batch * p.K;
        if (items > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
        const size_t ldsp = ((size_t)(5 + sp.m) * (sp.n + 1) * sp.d + 2 * ((size_t)sp.m * (sp.m + 2) + 1) * 4) * sizeof(double) + 64;  // tiles (odd column stride n + 1), two copies of the sums, counters
        if (!ctx->sp_fhess && ldsp <= (size_t)ctx->max_lds) {  // (the source is generated once per context)
            const std::string src = sparse_source(sp);
            const std::string key = "sparse:" + std::to_string(std::hash<std::string>{}(src));
            ctx->sp_fhess = jit_compile(ctx->device, key, src, "pcl_hess_sparse_kernel", true);
            if (ctx->sp_fhess) ctx->sp_fval = jit_compile(ctx->device, key, src, "pcl_sparse_values_kernel", true);
        }
        hipFunction_t fval = ctx->sp_fval, fmain = ctx->sp_fhess;
        if (fmain && fval) {
            if (ctx->sp_gvals_cap < (long long)ctx->desc.batch * p.K) {
                if (ctx->dsp_gvals) (void)hipFree(ctx->dsp_gvals);
                ctx->dsp_gvals = nullptr;
                ctx->sp_gvals_cap = 0;
                HIP_TRY(ctx, hipMalloc((void **)&ctx->dsp_gvals, ((size_t)ctx->desc.batch * p.K * sp.nzp + 32) * sizeof(double)));
                ctx->sp_gvals_cap = (long long)ctx->desc.batch * p.K;
            }
            {
                void *args[] = {(void *)&p, (void *)&ctx->dsp_pos, (void *)&ctx->dsp_coef, (void *)&ctx->dsp_gvals};
                HIP_TRY(ctx, hipModuleLaunchKernel(fval, (unsigned)items, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr));
            }
            const long long slots = std::max(ctx->n_cu, 1), rounds = (items + slots - 1) / slots;
            long long grid = (items + rounds - 1) / rounds;  // every workgroup walks the same number of intervals
            if (ctx->opt_grid > 0) grid = std::min<long long>(items, ctx->opt_grid);
            void *args[] = {(void *)&p, (void *)&ctx->dsp_gvals, (void *)&ctx->dsp_glv};
            HIP_TRY(ctx, hipModuleLaunchKernel(fmain, (unsigned)grid, 1, 1, 64 * (sp.m + 2), 1, 1, (unsigned)ldsp, ctx->stream, args, nullptr));
            ctx->last_hess_kernel = 6;  // the pattern-compiled kernel
            return PCL_OK;
        }
        if (ldsp > (size_t)ctx->max_lds)
            ctx->sp_hess_unfit = 1;  // (the residual kernel of the same source needs one tile per wave and stays available)
        else {
            ctx->sp_failed = 1;
            if (int rc = jit_fell_back(ctx, "Hessian of the Lagrangian (order 4)"); rc != PCL_ENOTIMPL) return rc;
        }
        if (ctx->opt_hess_kernel == 4)
            return fail(ctx, PCL_ESHAPE, "hess_kernel=4: the pattern-compiled kernel is not available (%zu B of LDS needed, %d available; %s)", ldsp, ctx->max_lds,
                        g_jit_note.c_str());
    } else if (ctx->opt_hess_kernel == 4) {
        return fail(ctx, PCL_ESHAPE, "hess_kernel=4 needs sparse iso generators (at most %d distinct drive magnitudes), a unitary problem with 9 <= d <= 32, 1..6 drives and jit=1", pcl_codegen::kMaxMags);
    }
    // version 3 (default where its tiles fit LDS): one workgroup per interval, jobs split by drive
    if (mf && (ctx->opt_hess_kernel == 0 || ctx->opt_hess_kernel == 3) && hess_v2_supported(ctx) && ctx->cols == ctx->desc.d && ctx->uell_w <= 2 &&
        ctx->n_upos <= 512 * PCL_NUE_H3 && (ctx->opt_hess_kernel == 3 || ctx->desc.d >= 12)) {
        const size_t lde = (p.d & 1) ? (size_t)p.n : (size_t)p.LD;
        const size_t nsc = (size_t)(p.m + 1) * (p.m + 2) / 2;
        const size_t lds3 = (lde * p.n + (8 + 2 * (size_t)p.m) * lde * 16 + 2 * (size_t)(p.m + 1) + 8 * nsc + 8) * sizeof(double) + 64;
        const bool st27 = ctx->opt_specialize && ctx->drives_antisym && p.d == 27 && p.m == 6;
        const bool st25 = ctx->opt_specialize && ctx->drives_antisym && p.d == 25 && p.m == 4;
        if (lds3 <= (size_t)ctx->max_lds) {
            const void *k3 = st27 ? (const void *)pcl_hess_kernel_v3<PCL_HESS_EW, 6, 27, true>
                           : st25 ? (const void *)pcl_hess_kernel_v3<PCL_HESS_EW, 4, 25, true> : nullptr;
            hipFunction_t j3 = nullptr;
            if (!k3 && ctx->opt_jit && ctx->opt_specialize) {  // any other shape: compiled on first use
                char inst[96];
                snprintf(inst, sizeof inst, "pcl_hess_kernel_v3<%d, %d, %d, %s>", PCL_HESS_EW, p.m, p.d, ctx->drives_antisym ? "true" : "false");
                j3 = jit_function(ctx->device, inst);
            }
            if (k3 || j3) {
                if (k3)
                    if (int rc = set_lds_attr(ctx, k3, 7, lds3)) return rc;
                const long long items = (long long)p.batch * p.K;
                if (items > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "too many work items");
                const long long slots = std::max(ctx->n_cu, 1), rounds = (items + slots - 1) / slots;
                long long grid = (items + rounds - 1) / rounds;  // every workgroup walks the same number of intervals
                if (ctx->opt_grid > 0) grid = std::min<long long>(items, ctx->opt_grid);
                void *args[] = {(void *)&p};
                if (j3)
                    HIP_TRY(ctx, hipModuleLaunchKernel(j3, (unsigned)grid, 1, 1, 512, 1, 1, (unsigned)lds3, ctx->stream, args, nullptr));
                else
                    HIP_TRY(ctx, hipLaunchKernel(k3, dim3((unsigned)grid), dim3(512), args, lds3, ctx->stream));
                HIP_TRY(ctx, hipGetLastError());
                ctx->last_hess_kernel = j3 ? 5 : 4;  // 4: version 3 (static instance), 5: version 3 compiled on first use
                return PCL_OK;
            }
        }
        if (ctx->opt_hess_kernel == 3) return fail(ctx, PCL_ESHAPE, "hess_kernel=3: no instance for this shape (tiles need %zu B of LDS, jit=%d)", lds3, (int)ctx->opt_jit);
    }
    if (ctx->opt_hess_kernel == 2 && !hess_v2_supported(ctx))
        return fail(ctx, PCL_ESHAPE, "hess_kernel=2 needs 1..6 drives with at most %d entries per row and column (have m=%d, widths %d/%d)",
                    PCL_HESS_EW, p.m, ctx->ell_w, ctx->ellt_w);
    if (mf && ctx->opt_hess_kernel != 1 && hess_v2_supported(ctx) && hess2_lds_bytes(p) <= (size_t)ctx->max_lds) {
        // one chunk of 16/(m+1) columns per wave and item when the interval's columns allow it, never more than 16 per slice
        const int ncw = 16 / (p.m + 1);
        int S = (p.cols + 4 * ncw - 1) / (4 * ncw);
        if (ctx->opt_cols_per_slice > 0) S = (p.cols + (int)std::min<int64_t>(ctx->opt_cols_per_slice, 16) - 1) / (int)std::min<int64_t>(ctx->opt_cols_per_slice, 16);
        S = std::max(S, (p.cols + 15) / 16);
        p.nc = (p.cols + S - 1) / S;
        p.S = (p.cols + p.nc - 1) / p.nc;
        const long long nbk = (long long)p.batch * p.K;
        const long long nbk_all = (long long)ctx->desc.batch * p.K;  // scratch is sized for every member, whatever the window
        if (p.S > 1 && !ctx->dhcnt) HIP_TRY(ctx, hipMalloc((void **)&ctx->dhcnt, (size_t)nbk_all * sizeof(unsigned int)));
        if (p.S > 1 && ctx->hpart_cap < nbk_all * p.S) {
            const size_t nscal = (size_t)(p.m + 1) * (p.m + 2) / 2;
            if (ctx->dhpart) (void)hipFree(ctx->dhpart);
            ctx->dhpart = nullptr;
            ctx->hpart_cap = 0;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->dhpart, (size_t)nbk_all * p.S * nscal * sizeof(double)));
            ctx->hpart_cap = nbk_all * p.S;
        }
        // arrival counters start from zero in every launch (a launch that faulted must not poison the next one)
        if (p.S > 1) HIP_TRY(ctx, hipMemsetAsync(ctx->dhcnt, 0, (size_t)nbk * sizeof(unsigned int), ctx->stream));
        p.hpart = ctx->dhpart;
        p.hcnt = ctx->dhcnt;
        const size_t lds = hess2_lds_bytes(p);
        hipFunction_t jith = nullptr;
        const bool hess_static = ctx->opt_specialize && ctx->drives_antisym && ((p.d == 27 && p.m == 6) || (p.d == 25 && p.m == 4));
        if (!hess_static && ctx->opt_jit && ctx->opt_specialize && ctx->cols == ctx->desc.d && ctx->ell_w >= 1 && p.d >= 12) {  // small d: launch-bound either way
            char inst[96];
            snprintf(inst, sizeof inst, "pcl_hess_kernel_v2<%d, %d, %d, %s>", PCL_HESS_EW, p.m, p.d, ctx->drives_antisym ? "true" : "false");
            jith = jit_function(ctx->device, inst);
        }
        const void *kern = ctx->drives_antisym ? hess_v2_kernel<PCL_HESS_EW, true>(p.m) : hess_v2_kernel<PCL_HESS_EW, false>(p.m);
        if (ctx->opt_specialize && p.d == 27 && p.m == 6 && ctx->drives_antisym)
            kern = (const void *)pcl_hess_kernel_v2<PCL_HESS_EW, 6, 27, true>;  // BASELINE config 3's shape
        else if (ctx->opt_specialize && p.d == 25 && p.m == 4 && ctx->drives_antisym)
            kern = (const void *)pcl_hess_kernel_v2<PCL_HESS_EW, 4, 25, true>;  // two 5-level transmons
        if (!jith)
            if (int rc = set_lds_attr(ctx, kern, 7, lds)) return rc;
        const long long items = nbk * p.S;
        const int per_cu = std::max(1, std::min(2, (int)((size_t)ctx->max_lds / lds)));
        long long grid = std::min<long long>(items, (long long)per_cu * ctx->n_cu);
        if (ctx->opt_grid > 0) grid = std::min<long long>(items, ctx->opt_grid);
        void *args[] = {(void *)&p};
        if (jith)
            HIP_TRY(ctx, hipModuleLaunchKernel(jith, (unsigned)grid, 1, 1, 256, 1, 1, (unsigned)lds, ctx->stream, args, nullptr));
        else
            HIP_TRY(ctx, hipLaunchKernel(kern, dim3((unsigned)grid), dim3(256), args, lds, ctx->stream));
        HIP_TRY(ctx, hipGetLastError());
        ctx->last_hess_kernel = jith ? 3 : 2;
        return PCL_OK;
    }
    // column chunk: as many columns as fit in half the LDS (two workgroups per CU)
    p.nc = p.cols;
    while (p.nc > 1 && hess_lds_bytes(p) > (size_t)ctx->max_lds / 2) p.nc = (p.nc + 1) / 2;
    const size_t lds = hess_lds_bytes(p);
    if (lds > (size_t)ctx->max_lds)
        return fail(ctx, PCL_ESHAPE, "Hessian kernel needs %zu B of LDS (> %d) for d=%d, m=%d", lds, ctx->max_lds, p.d, p.m);
    const long long grid = (long long)p.batch * p.K;
    auto kern = mf ? pcl_hess_kernel<true> : pcl_hess_kernel<false>;
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_hess_kernel = 1;
    return PCL_OK;
}

// --- order policy ---------------------------------------------------------------------------------
// The reference's constraint is x_{k+1} = exp(dt_k G(u_k)) x_k (docs/src/concepts/index.md:21); the diagonal Pade residual of order 2q
// deviates from it by  kappa_q theta^(2q+1) (1 + O(theta^2)) |x|,  theta = |dt_k G(u_k)|_2,  kappa_q = (q!)^2 / ((2q)! (2q+1)!)  -- the
// leading term of exp - r_qq.  (Measured on exp-feasible trajectories, scripts/pade_vs_exp.py -> profiles/pade_vs_exp.json: config 3, theta = 0.438:
// 1.6e-5 / 1.6e-11 / 7.9e-15 at orders 4 / 8 / 10 against 2.2e-5 / 2.3e-11 / 1.1e-14 from this bound.)  pade_order = 0 asks for the
// smallest order whose bound is below a tolerance, theta from the problem's bounds (pcl_set_order_policy) or, failing that, from the
// first trajectory a host-pointer entry point sees (with a margin of 1.5 on theta: later iterates move inside their bounds).
static double spectral_norm(const double *A, int n) {
    // Power iteration on A^T A (n <= 64: microseconds) from two generic, non-symmetric start vectors -- a symmetric one (all ones) is
    // annihilated by generators whose rows sum to zero and can sit orthogonal to the top singular vector -- the larger result counts; the
    // guaranteed bound sqrt(|A|_1 |A|_inf) >= |A|_2 stands in when the iteration has nothing to say and caps a result that exceeds it.
    double n1 = 0.0, ninf = 0.0;
    {
        std::vector<double> rs(n, 0.0);
        for (int j = 0; j < n; ++j) {
            double cs = 0.0;
            for (int i = 0; i < n; ++i) cs += std::fabs(A[i + (size_t)n * j]), rs[i] += std::fabs(A[i + (size_t)n * j]);
            n1 = std::max(n1, cs);
        }
        for (int i = 0; i < n; ++i) ninf = std::max(ninf, rs[i]);
    }
    const double upper = std::sqrt(n1 * ninf);
    if (upper == 0.0) return 0.0;
    std::vector<double> x(n), y(n), z(n);
    double best = 0.0;
    for (int seed = 0; seed < 2; ++seed) {
        for (int j = 0; j < n; ++j) x[j] = 1.0 + 0.61803398875 * std::sin(1.0 + (seed ? 2.39996323 : 1.0) * (j + 1)) + (seed ? 0.25 * ((j * 7 + 3) % 5) : 0.0);
        double s = 0.0;
        bool converged = false;
        for (int it = 0; it < 200 && !converged; ++it) {
            for (int i = 0; i < n; ++i) {
                double a = 0.0;
                for (int j = 0; j < n; ++j) a += A[i + (size_t)n * j] * x[j];
                y[i] = a;
            }
            for (int j = 0; j < n; ++j) {
                double a = 0.0;
                for (int i = 0; i < n; ++i) a += A[i + (size_t)n * j] * y[i];
                z[j] = a;
            }
            double nz = 0.0, nx = 0.0;
            for (int j = 0; j < n; ++j) nz += z[j] * z[j], nx += x[j] * x[j];
            if (nz == 0.0) {
                s = 0.0;
                break;
            }
            const double s1 = std::sqrt(std::sqrt(nz) / std::sqrt(nx));
            nz = std::sqrt(nz);
            for (int j = 0; j < n; ++j) x[j] = z[j] / nz;
            converged = it >= 8 && std::fabs(s1 - s) <= 1e-12 * s1;  // (a plateau in the first steps may be a sub-dominant singular value)
            s = s1;
        }
        best = std::max(best, converged ? s : s * 1.01);  // (not converged to round-off: a hair on the safe side)
    }
    if (best == 0.0) return upper;
    return std::min(best, upper);
}
// the smallest diagonal Pade order whose bound is below tol; *met (when given) says whether any order up to 10 is
static int order_for(double theta, double tol, bool *met = nullptr) {
    double fact[12];
    fact[0] = 1.0;
    for (int i = 1; i < 12; ++i) fact[i] = fact[i - 1] * i;
    if (met) *met = true;
    for (int q = 1; q <= 5; ++q) {
        const double kappa = fact[q] * fact[q] / (fact[2 * q] * fact[2 * q + 1]);
        if (kappa * std::pow(theta, 2 * q + 1) <= tol) return 2 * q;
    }
    if (met) *met = false;
    return 10;
}
// (order 10 chosen although its bound exceeds the tolerance: said in pcl_last_error's text -- no error code -- and readable as option order_tol_met)
static void note_order(pcl_ctx *ctx, double theta, bool met) {
    ctx->order_tol_met = met ? 1 : 0;
    if (!met) {
        char buf[256];
        snprintf(buf, sizeof buf, "order policy: |dt G| <= %.3g is too large for any diagonal Pade order up to 10 to stay within %.3g of the exp constraint; order 10 is used (option order_tol_met = 0)", theta, ctx->order_tol);
        ctx->err = buf;
        if (getenv("PCL_VERBOSE")) fprintf(stderr, "piccolo_hip: %s\n", buf);
    }
}
static void set_order(pcl_ctx *ctx, int order, double theta) {
    if (ctx->desc.pade_order != order) {  // (modules are per order: the handles of the previous one are dropped, the modules stay cached)
        ctx->v4_f = ctx->v4_ft = ctx->v4_feval = ctx->v4_fevalc = ctx->v4_fhess = ctx->v4_fhess2 = ctx->v4_fhessc = ctx->v4_fhessp = nullptr;
        ctx->v4_failed = ctx->v4_hess_failed = ctx->v4_hessc_failed = ctx->v4_ft_failed = 0;
#ifdef PCL_LAB
        ctx->res.f = nullptr;  // (the resident module bakes the order in as well)
#endif
    }
    ctx->desc.pade_order = order;
    ctx->order_theta = theta;
}
// theta = dt_max max_{|u_l| <= u_max_l} |G_drift + sum_l u_l G_l|_2 over n_g0 drifts (n x n, column-major; any real generators).  The norm is convex
// in u, so the maximum over the box sits at a vertex: all 2^m sign patterns are tried (m <= 10; 64 norms of a 54 x 54 matrix at config 3: tens of
// ms, once per context).  The triangle inequality |G_drift| + sum_l u_max_l |G_l| -- what rounds 3-4 used, and the fallback for m > 10 -- overshoots
// by half where the drives act on different subsystems (config 3: 10.5 against 6.86) and would ask for an order that no trajectory in the box needs.
// Further drifts (ensemble members): |G_drift_b + D| <= max_vertex |G_drift_0 + D| + |G_drift_b - G_drift_0|.
static double policy_theta(int n, int m, const double *G0, size_t n_g0, const double *Gj, double dt_max, const double *u_max) {
    const size_t nn = (size_t)n * n;
    double gd = 0.0;
    for (int l = 0; l < m; ++l) gd += std::fabs(u_max[l]) * spectral_norm(Gj + (size_t)l * nn, n);
    const double tri0 = spectral_norm(G0, n) + gd;
    double vmax = tri0;
    if (m <= 10) {
        std::vector<double> G(nn);
        vmax = 0.0;
        for (unsigned sgn = 0; sgn < (1u << m); ++sgn) {
            for (size_t e = 0; e < nn; ++e) {
                double a = G0[e];
                for (int l = 0; l < m; ++l) a += ((sgn >> l) & 1u ? -1.0 : 1.0) * std::fabs(u_max[l]) * Gj[(size_t)l * nn + e];
                G[e] = a;
            }
            vmax = std::max(vmax, spectral_norm(G.data(), n));
        }
        vmax = std::min(vmax, tri0);
    }
    double theta = dt_max * vmax;
    std::vector<double> Dm(nn);
    for (size_t b = 1; b < n_g0; ++b) {
        for (size_t e = 0; e < nn; ++e) Dm[e] = G0[b * nn + e] - G0[e];
        theta = std::max(theta, dt_max * (vmax + spectral_norm(Dm.data(), n)));
    }
    return theta;
}
// The policy without a context or a device (what pcl_set_order_policy applies; the CPU tests pin it): n = generator dimension (2 d for unitary / ket
// problems), n_g0 drifts.  order_out: the smallest diagonal Pade order whose bound kappa_q theta^(2q+1) is <= tol (10 when none is: *met_out = 0).
extern "C" int pcl_order_for_bounds(int32_t n, int32_t m, const double *G0, int32_t n_g0, const double *Gj, double dt_max, const double *u_max, double tol,
                                    double *theta_out, int32_t *order_out, int32_t *met_out) {
    if (n < 1 || n > 4096 || m < 0 || n_g0 < 1 || !G0 || (m > 0 && (!Gj || !u_max)) || !(dt_max > 0.0) || !(tol > 0.0)) return PCL_EINVAL;
    const double theta = policy_theta(n, m, G0, (size_t)n_g0, Gj, dt_max, u_max);
    bool met = true;
    const int order = order_for(theta, tol, &met);
    if (theta_out) *theta_out = theta;
    if (order_out) *order_out = order;
    if (met_out) *met_out = met ? 1 : 0;
    return PCL_OK;
}
extern "C" int pcl_set_order_policy(pcl_ctx *ctx, double dt_max, const double *u_max, double tol, int32_t *order_out) {
    if (!ctx) return PCL_EINVAL;
    if (!(dt_max > 0.0) || !(tol > 0.0) || (ctx->desc.n_drives > 0 && !u_max)) return fail(ctx, PCL_EINVAL, "pcl_set_order_policy: need dt_max > 0, tol > 0 and the drives' bounds");
#ifdef PCL_LAB
    if (ctx->res.active) return fail(ctx, PCL_EINVAL, "pcl_set_order_policy: a resident evaluator is running (pcl_resident_stop first)");
#endif
    const int n = ctx->n, m = ctx->desc.n_drives;
    const double theta = policy_theta(n, m, ctx->hG0.data(), ctx->hG0.size() / ((size_t)n * n), ctx->hGj.data(), dt_max, u_max);
    ctx->order_tol = tol;
    bool met = true;
    set_order(ctx, order_for(theta, tol, &met), theta);
    note_order(ctx, theta, met);
    if (order_out) *order_out = ctx->desc.pade_order;
    return PCL_OK;
}
// pade_order = 0 and no policy yet: the first trajectory on the host decides (host-pointer entry points); device-pointer entry points
// have nothing to look at
static int resolve_order(pcl_ctx *ctx, const double *Z_host, const char *where) {
    if (ctx->desc.pade_order != 0) return PCL_OK;
    if (!Z_host) return fail(ctx, PCL_EINVAL, "%s: the context was created with pade_order = 0; call pcl_set_order_policy (or a host-pointer entry point) first", where);
    const pcl_desc &D = ctx->desc;
    const int n = ctx->n, m = D.n_drives;
    const int nbuf = D.batch_mode == PCL_BATCH_TRAJ ? D.batch : 1;
    std::vector<double> G((size_t)n * n);
    double theta = 0.0;
    for (size_t g0 = 0; g0 * n * n < ctx->hG0.size(); ++g0)
        for (int b = 0; b < nbuf; ++b) {
            // (a multistart constructed from ONE trajectory tiled `batch` times: one pass of norms, not `batch`)
            if (b > 0 && !memcmp(Z_host + (size_t)b * D.N * D.z_dim, Z_host, (size_t)D.N * D.z_dim * sizeof(double))) continue;
            for (int k = 0; k + 1 < D.N; ++k) {
                const double *z = Z_host + ((size_t)b * D.N + k) * D.z_dim;
                for (size_t e = 0; e < (size_t)n * n; ++e) {
                    double a = ctx->hG0[g0 * n * n + e];
                    for (int l = 0; l < m; ++l) a += z[D.u_off + l] * ctx->hGj[(size_t)l * n * n + e];
                    G[e] = a;
                }
                theta = std::max(theta, std::fabs(z[D.dt_off]) * spectral_norm(G.data(), n));
            }
        }
    bool met = true;
    set_order(ctx, order_for(1.5 * theta, ctx->order_tol, &met), 1.5 * theta);
    note_order(ctx, 1.5 * theta, met);
    return PCL_OK;
}

extern "C" int pcl_set_order_from_trajectory(pcl_ctx *ctx, const double *Z_host, double tol, int32_t *order_out) {
    if (!ctx) return PCL_EINVAL;
    if (!Z_host) return fail(ctx, PCL_EINVAL, "pcl_set_order_from_trajectory: NULL trajectory");
#ifdef PCL_LAB
    if (ctx->res.active) return fail(ctx, PCL_EINVAL, "pcl_set_order_from_trajectory: a resident evaluator is running (pcl_resident_stop first)");
#endif
    if (tol > 0.0) ctx->order_tol = tol;
    const int keep = ctx->desc.pade_order;
    ctx->desc.pade_order = 0;  // (resolve_order decides for contexts without an order; set_order drops the previous order's module handles)
    const int rc = resolve_order(ctx, Z_host, "pcl_set_order_from_trajectory");
    if (rc != PCL_OK) {
        ctx->desc.pade_order = keep;
        return rc;
    }
    if (order_out) *order_out = ctx->desc.pade_order;
    return PCL_OK;
}

// --- device-pointer API -----------------------------------------------------------------------
// A launch with slice tickets leaves its counters zero only when it has finished, and so do the Hessian kernels that exchange rows of sums through
// per-context arrival counters (column groups, two workgroups per interval): a launch on ANOTHER stream must not start before that (launches on
// one stream are ordered anyway; the counters' first-use memset is ordered on the stream of that moment too).  Switching streams therefore waits
// for the old one if such a launch may be in flight.
static int change_stream(pcl_ctx *ctx, hipStream_t s) {
    if (s != ctx->stream && ctx->ticket_launched) {
        ON_DEVICE(ctx);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->ticket_launched = false;
    }
    ctx->stream = s;
    ctx->tickets_dirty = true;
    return PCL_OK;
}
extern "C" int pcl_set_stream(pcl_ctx *ctx, void *s) {
    if (!ctx) return PCL_EINVAL;
    return change_stream(ctx, (hipStream_t)s);  // NULL is HIP's legacy default stream
}
extern "C" int pcl_reset_stream(pcl_ctx *ctx) {
    if (!ctx) return PCL_EINVAL;
    return change_stream(ctx, ctx->own_stream);
}
extern "C" int pcl_sync(pcl_ctx *ctx) {
    if (!ctx) return PCL_EINVAL;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx, "pcl_sync");
}
extern "C" int pcl_eval_dev(pcl_ctx *ctx, const double *Z, double *delta) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !delta) return fail(ctx, PCL_EINVAL, "pcl_eval_dev: NULL pointer");
    return launch_fused(ctx, Z, delta, nullptr, false);
}
extern "C" int pcl_eval_jac_dev(pcl_ctx *ctx, const double *Z, double *delta, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !vals) return fail(ctx, PCL_EINVAL, "pcl_eval_jac_dev: NULL pointer");
    return launch_fused(ctx, Z, delta, vals, false);
}

#ifdef PCL_LAB
#include "pcl_host_resident.hpp"  // pcl_resident_* (include/piccolo_hip_lab.h): measured, loses, kept for the record
#endif

extern "C" int pcl_jac_dev(pcl_ctx *ctx, const double *Z, double *vals) {  // eval_jacobian alone: no residual is written
    if (!ctx) return PCL_EINVAL;
    if (!Z || !vals) return fail(ctx, PCL_EINVAL, "pcl_jac_dev: NULL pointer");
    return launch_fused(ctx, Z, nullptr, vals, false);
}
extern "C" int pcl_eval_jac_compact_dev(pcl_ctx *ctx, const double *Z, double *delta, double *compact) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !compact) return fail(ctx, PCL_EINVAL, "pcl_eval_jac_compact_dev: NULL pointer");
    return launch_fused(ctx, Z, delta, compact, true);
}
extern "C" int pcl_jac_expand_dev(pcl_ctx *ctx, const double *compact, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!compact || !vals) return fail(ctx, PCL_EINVAL, "pcl_jac_expand_dev: NULL pointer");
    ON_DEVICE(ctx);
    const long long n_bk = (long long)ctx->win_count * ctx->K;
    if (ctx->cols == 1) {  // one state column (kets, compact density vectors): the compact layout IS the full layout
        HIP_TRY(ctx, hipMemcpyAsync(vals, compact, (size_t)(n_bk * jac_per_full(ctx)) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        return PCL_OK;
    }
    // slices of 3 state columns (70 KB per workgroup at config 3), the option cols_per_slice overrides
    const int cpi = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_cols_per_slice > 0 ? ctx->opt_cols_per_slice : 3, ctx->cols));
    const long long tail2 = ((long long)ctx->x_dim * (ctx->desc.n_drives + 1)) >> 1;
    const long long per_bk = 2LL * ((ctx->cols + cpi - 1) / cpi) + (tail2 + 2047) / 2048;
    const long long grid = n_bk * per_bk;
    if (grid > 0x7fffffffLL) return fail(ctx, PCL_ESHAPE, "pcl_jac_expand_dev: %lld work items exceed the grid limit", grid);
    if ((ctx->n & 1) || ((long long)ctx->x_dim * (ctx->desc.n_drives + 1)) % 2) return fail(ctx, PCL_ESHAPE, "pcl_jac_expand_dev: odd block sizes");
    hipLaunchKernelGGL(pcl_expand_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, compact, vals, ctx->cols, ctx->n,
                       ctx->desc.n_drives, n_bk, cpi, (int)(ctx->opt_nt > 0 ? ctx->opt_nt : 0));
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
extern "C" int pcl_hess_dev(pcl_ctx *ctx, const double *Z, const double *mu, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !mu || !vals) return fail(ctx, PCL_EINVAL, "pcl_hess_dev: NULL pointer");
    return launch_hess(ctx, Z, mu, vals);
}

extern "C" int pcl_rollout_dev(pcl_ctx *ctx, const double *Z, double *X_out) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !X_out) return fail(ctx, PCL_EINVAL, "pcl_rollout_dev: NULL pointer");
    ON_DEVICE(ctx);
    KParams p;
    fill_params(ctx, p);
    p.Z = Z + (ctx->desc.batch_mode == PCL_BATCH_TRAJ ? (long long)ctx->win_first * ctx->desc.z_dim * ctx->desc.N : 0);
    p.xout = X_out;
    if (!ctx->dexpm) HIP_TRY(ctx, hipMalloc((void **)&ctx->dexpm, (size_t)ctx->desc.batch * p.K * p.n * p.n * sizeof(double)));
    p.expm = ctx->dexpm;
    const size_t lds_a = (3 * (size_t)p.LD * p.n + 8 + p.m + 64) * sizeof(double);
    const size_t lds_b = ((size_t)p.LD * p.n + 2 * (size_t)p.LD * p.cols) * sizeof(double);
    if (lds_a > (size_t)ctx->max_lds || lds_b > (size_t)ctx->max_lds) return fail(ctx, PCL_ESHAPE, "pcl_rollout_dev: tiles exceed LDS");
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)pcl_expm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)pcl_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    hipLaunchKernelGGL(pcl_expm_kernel, dim3((unsigned)((long long)p.batch * p.K)), dim3(256), lds_a, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(pcl_chain_kernel, dim3((unsigned)p.batch), dim3(256), lds_b, ctx->stream, p);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}

// --- host-pointer API (staging buffers owned by the context) --------------------------------------
static int ensure(pcl_ctx *ctx, double **buf, long long count) {
    if (*buf) return PCL_OK;
    hipError_t e = hipMalloc((void **)buf, (size_t)count * sizeof(double));
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? PCL_ENOMEM : PCL_EHIP, "hipMalloc(%lld doubles): %s", count, hipGetErrorString(e));
    return PCL_OK;
}
#define TRY(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != PCL_OK) return rc_; \
    } while (0)

static int ensure_pinned(pcl_ctx *ctx, double **buf, long long count) {
    if (*buf) return PCL_OK;
    hipError_t e = hipHostMalloc((void **)buf, (size_t)count * sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, PCL_ENOMEM, "hipHostMalloc(%lld doubles): %s", count, hipGetErrorString(e));
    return PCL_OK;
}
// Thread count of the host expansion.  Default (0): min(cores / 2, 32) -- the path is bound by the host's memory writes into the caller's
// pageable array, 16 ... 64 threads deliver 1.0-1.3 k evaluations per second on the boxes of the pool, which count is best changes from box to
// box and from run to run (page placement of the caller's array, neighbours).  host_threads = -1 asks for a sweep: the context's first 2 x 6
// host-delivered calls are each run at one of six candidate counts -- real calls, real results -- and the count whose fastest call was
// shortest serves the context from then on (measured: 1.6 k on one box, no better than the default on two others; hence opt-in).
static const int kHostTuneCand[] = {16, 24, 32, 48, 64, 96};
static const int kHostTuneN = 6;
static int host_threads(const pcl_ctx *ctx) {
    if (ctx->opt_host_threads > 0) return (int)std::min<int64_t>(ctx->opt_host_threads, 256);
    if (ctx->host_threads_tuned > 0) return ctx->host_threads_tuned;
    const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
    if (ctx->opt_host_threads < 0 && ctx->host_tune_calls < 2 * kHostTuneN) return (int)std::min<unsigned>((unsigned)kHostTuneCand[ctx->host_tune_calls % kHostTuneN], hw);
    // ... and never more than the CPUs the cgroup grants (cpu.max): a container that sees 256 hardware threads and is allowed 16 CPUs delivered
    // 1,390 evaluations/s sustained with 16 threads and 650-1,000 with 32 (1,570 per median call: the team wins single calls and is throttled
    // over a run -- round 5, bench.py other_rates.host_delivered.paths)
    unsigned def = std::max(1u, std::min(hw / 2, 32u));
    static const double quota = pcl_host::cgroup_quota_cpus();
    if (quota >= 1.0) def = std::min(def, (unsigned)quota);
    return (int)std::max(1u, def);
}

// Host-pointer evaluation.  Two ways to deliver the Jacobian values into the caller's (pageable) array:
//   full    the kernel writes the full triplet-order values in HBM and 132.8 MB (config 3) cross PCIe;
//   compact the kernel writes only the unique tiles (12.7 MB), they cross PCIe into pinned staging in chunks of intervals,
//           and the host's threads replicate them into the caller's array with streaming stores while the next chunk is
//           in flight -- bound by host memory write bandwidth instead of by PCIe.  Default whenever blocks are replicated.
static int host_eval_jac(pcl_ctx *ctx, const double *Z, double *delta, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || (!delta && !vals)) return fail(ctx, PCL_EINVAL, "NULL pointer");
    ON_DEVICE(ctx);
    TRY(resolve_order(ctx, Z, "pcl_eval_jac"));
    const auto t_begin = std::chrono::steady_clock::now();
    const long long nbk = (long long)ctx->win_count * ctx->K, nbk_all = (long long)ctx->desc.batch * ctx->K;
    const long long nv = jac_per_full(ctx) * nbk;
    const bool compact_path = vals && ctx->cols > 1 && ctx->opt_host_path != 1;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    TRY(ensure(ctx, &ctx->ddelta, n_rows_all(ctx)));
    TRY(ensure_pinned(ctx, &ctx->hZ, z_len(ctx)));
    memcpy(ctx->hZ, Z, (size_t)z_len(ctx) * sizeof(double));  // pinned source: the H2D copy is one asynchronous DMA
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dZ, ctx->hZ, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (!compact_path) {
        if (vals) TRY(ensure(ctx, &ctx->dvals, jac_per_full(ctx) * nbk_all));
        TRY(launch_fused(ctx, ctx->dZ, ctx->ddelta, vals ? ctx->dvals : nullptr, false));
        if (delta) HIP_TRY(ctx, hipMemcpyAsync(delta, ctx->ddelta, n_rows(ctx) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (vals) HIP_TRY(ctx, hipMemcpyAsync(vals, ctx->dvals, nv * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return check_device_error(ctx, "pcl_eval_jac");
    }
    const long long cper = jac_per_compact(ctx), fper = jac_per_full(ctx);
    TRY(ensure(ctx, &ctx->dcomp_host, cper * nbk_all));
    TRY(ensure_pinned(ctx, &ctx->hcompact, cper * nbk_all));
    TRY(ensure_pinned(ctx, &ctx->hdelta, n_rows_all(ctx)));
    const int n_chunks = (int)std::max<long long>(1, std::min<long long>(std::min<int64_t>(ctx->opt_host_chunks, 8), nbk));
    for (int c = 0; c < n_chunks; ++c)
        if (!ctx->ev_chunk[c]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_chunk[c], hipEventDisableTiming));
    TRY(launch_fused(ctx, ctx->dZ, ctx->ddelta, ctx->dcomp_host, true));
    for (int c = 0; c < n_chunks; ++c) {
        const long long lo = nbk * c / n_chunks, hi = nbk * (c + 1) / n_chunks;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->hcompact + lo * cper, ctx->dcomp_host + lo * cper, (size_t)((hi - lo) * cper) * sizeof(double),
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_chunk[c], ctx->stream));
    }
    if (delta) HIP_TRY(ctx, hipMemcpyAsync(ctx->hdelta, ctx->ddelta, n_rows(ctx) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    const bool tuning = ctx->opt_host_threads < 0 && ctx->host_threads_tuned == 0 && ctx->host_tune_calls < 2 * kHostTuneN && nv * 8 >= (32LL << 20);
    const int want_threads = host_threads(ctx);
    if (!ctx->pool || ctx->pool->size() != want_threads - 1) {
        delete ctx->pool;
        ctx->pool = new (std::nothrow) pcl_host::Pool(want_threads - 1);  // the calling thread is the last worker
        if (!ctx->pool) return fail(ctx, PCL_ENOMEM, "thread pool");
    }
    const int cols = ctx->cols;
    const long long nn = (long long)ctx->n * ctx->n, tail = ctx->x_dim * (ctx->desc.n_drives + 1);
    const double *hc = ctx->hcompact;
    int store_w = 0;
    const pcl_host::stream_copy_fn copy = pcl_host::pick_stream_copy((int)ctx->opt_host_store_bytes, &store_w);
    ctx->last_host_store_bytes = store_w;
    ctx->pool->begin(2 * nbk, [=](long long job) {
        const long long bk = job >> 1;
        pcl_host::expand_interval(vals + bk * fper, hc + bk * cper, cols, nn, tail, (int)(job & 1), copy);
    });
    int rc = PCL_OK;
    for (int c = 0; c < n_chunks; ++c) {
        if (rc == PCL_OK && hipEventSynchronize(ctx->ev_chunk[c]) != hipSuccess) rc = PCL_EHIP;
        ctx->pool->publish(2 * (nbk * (c + 1) / n_chunks));  // (on an error the workers still run to completion, on stale bytes)
    }
    ctx->pool->wait();
    if (rc != PCL_OK) return fail(ctx, PCL_EHIP, "hipEventSynchronize: %s", hipGetErrorString(hipGetLastError()));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (delta) memcpy(delta, ctx->hdelta, (size_t)n_rows(ctx) * sizeof(double));
    if (tuning) {  // this call was one sample of the sweep
        const double t_call = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        const int ci = ctx->host_tune_calls % kHostTuneN;
        ctx->host_tune_t[ci] = std::min(ctx->host_tune_t[ci], t_call);
        if (++ctx->host_tune_calls == 2 * kHostTuneN) {
            int bi = 0;
            for (int c = 1; c < kHostTuneN; ++c)
                if (ctx->host_tune_t[c] < ctx->host_tune_t[bi]) bi = c;
            ctx->host_threads_tuned = std::min(kHostTuneCand[bi], (int)std::max(2u, std::thread::hardware_concurrency()));
            ctx->host_expand_GBps = (double)nv * 8.0 / ctx->host_tune_t[bi] / 1e9;
        }
    }
    return check_device_error(ctx, "pcl_eval_jac");
}
extern "C" int pcl_eval(pcl_ctx *ctx, const double *Z, double *delta) {
    if (ctx && !delta) return fail(ctx, PCL_EINVAL, "pcl_eval: delta is NULL");
    return host_eval_jac(ctx, Z, delta, nullptr);
}
extern "C" int pcl_jac(pcl_ctx *ctx, const double *Z, double *vals) {
    if (ctx && !vals) return fail(ctx, PCL_EINVAL, "pcl_jac: vals is NULL");
    return host_eval_jac(ctx, Z, nullptr, vals);
}
extern "C" int pcl_eval_jac(pcl_ctx *ctx, const double *Z, double *delta, double *vals) {
    if (ctx && (!delta || !vals)) return fail(ctx, PCL_EINVAL, "pcl_eval_jac: NULL output");
    return host_eval_jac(ctx, Z, delta, vals);
}
extern "C" int pcl_hess(pcl_ctx *ctx, const double *Z, const double *mu, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !mu || !vals) return fail(ctx, PCL_EINVAL, "pcl_hess: NULL pointer");
    ON_DEVICE(ctx);
    TRY(resolve_order(ctx, Z, "pcl_hess"));
    const long long nv = hess_per(ctx) * ctx->win_count * ctx->K;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    TRY(ensure(ctx, &ctx->dmu, n_rows_all(ctx)));
    TRY(ensure(ctx, &ctx->dhess, hess_per(ctx) * ctx->desc.batch * ctx->K));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dmu, mu, n_rows(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TRY(launch_hess(ctx, ctx->dZ, ctx->dmu, ctx->dhess));
    HIP_TRY(ctx, hipMemcpyAsync(vals, ctx->dhess, nv * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx, "pcl_hess");
}

extern "C" int pcl_rollout(pcl_ctx *ctx, const double *Z, double *X_out) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !X_out) return fail(ctx, PCL_EINVAL, "pcl_rollout: NULL pointer");
    ON_DEVICE(ctx);
    const long long nv = (long long)ctx->win_count * ctx->desc.N * ctx->x_dim;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    TRY(ensure(ctx, &ctx->dxout, (long long)ctx->desc.batch * ctx->desc.N * ctx->x_dim));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TRY(pcl_rollout_dev(ctx, ctx->dZ, ctx->dxout));
    HIP_TRY(ctx, hipMemcpyAsync(X_out, ctx->dxout, nv * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}

#include "pcl_host_objective.hpp"
#include "pcl_host_comm.hpp"
#include "pcl_host_options.hpp"

/*
 * piccolo_hip.h -- C ABI of libpiccolo_hip.so: the MI355X (gfx950) evaluator for
 * the Pade-integrator collocation constraint of Piccolo.jl / DirectTrajOpt.jl.
 *
 * This is the drop-in boundary for ONE path of the reference: what a
 * `DirectTrajOpt.AbstractIntegrator` has to provide to the NLP evaluator.  Each
 * entry point names the reference interface it replaces (paths relative to the
 * reference checkout, harmoniqs/Piccolo.jl v2.0.2; [EXT] = un-vendored
 * DirectTrajOpt.jl, known from Piccolo's call sites):
 *
 *   pcl_create            BilinearIntegrator(qtraj::UnitaryTrajectory, N)     src/control/integrators.jl:35-51
 *                         BilinearIntegrator(qtraj::KetTrajectory, N)         src/control/integrators.jl:58-74  (state_cols = 1)
 *                         BilinearIntegrator(qtraj::SamplingTrajectory, N)    src/control/integrators.jl:134-162
 *                         (copies G_drift / G_drives of sys.G, quantum_systems.jl:225-226,
 *                          composite_quantum_systems.jl:124-132, and the component ranges of the
 *                          NamedTrajectory, named_trajectory_conversion.jl:339-351)
 *   pcl_constraint_dim    B.dim == x_dim*(N-1), B.x_dim                        src/control/integrators.jl:307-309
 *   pcl_eval[_dev]        evaluate!(delta, B, traj) [EXT]                      src/control/integrators.jl:311,777
 *   pcl_jac_nnz/structure jacobian structure handed to MOI [EXT]; shape pin    src/control/integrators.jl:780-783
 *   pcl_jac[_dev]         eval_jacobian(B, traj) [EXT]                         src/control/integrators.jl:780
 *   pcl_eval_jac[_dev]    eval_constraint + eval_constraint_jacobian of one IPM iteration (fused)
 *   pcl_hess_nnz/structure, pcl_hess[_dev]
 *                         hessian_structure / Hessian-of-Lagrangian [EXT]      test/aqua.jl:6-9,
 *                                                                              src/control/templates/spline_pulse_problem.jl:96
 *
 * Semantics.  With h = dt_k, G = G0 + sum_j u_{k,j} G_j (n x n, n = 2d), X_k the
 * n x d matrix whose column c is Utilde_k[c*n .. (c+1)*n) (isomorphisms.jl:110-118):
 *     delta_k = B^-(hG) X_{k+1} - B^+(hG) X_k ,   B^{+-}(A) = I +- A/2 + A^2/12      (Pade order 4)
 * (BASELINE.json north_star; the reference's own constraint is the matrix
 * exponential x_{k+1} = exp(h Ghat) x_k, docs/src/concepts/index.md:21, of which
 * this is the (2,2) diagonal Pade discretisation -- see DESIGN.md.)
 *
 * Variable vector = [traj.datavec ; traj.global_data]; component i of knot k is
 * variable k*z_dim + i (integrators.jl:781-783).  Rows: member b, interval k,
 * component r -> b*x_dim*K + k*x_dim + r (integrators are concatenated in
 * prob.integrators order, integrators.jl:316-317); the caller adds the row offset
 * of this integrator block inside the full constraint vector.
 *
 * Jacobian values, per (member b, interval k), contiguous block of
 * jac_nnz_per_interval = 2*d*n*n + x_dim*(m+1) doubles, blocks ordered b-major then k:
 *     seg 0  d delta/d X_k      for c<d, j<n, i<n : -B^+[i,j]   row c*n+i, col x_off + c*n+j     (knot k)
 *     seg 1  d delta/d X_{k+1}  same loop          :  B^-[i,j]                                    (knot k+1)
 *     tail   for c<cols (state column): for l<m, i<n : d delta[c*n+i]/d u_l   col u_off + l         (knot k)
 *                                       then   i<n : d delta[c*n+i]/d dt    col dt_off
 *            (column-major: the (m+1)*n doubles one state column produces are contiguous -- full-line stores)
 * Hessian-of-Lagrangian values per (b,k), hess_nnz_per_interval = (m+1)(m+2)/2 + 2*x_dim*(m+1):
 *     seg 0 (u_i,u_j) j<=i | seg 1 (dt,u_j) | seg 2 (dt,dt) | seg 3 (u_l, X_k[r]) | seg 4 (dt, X_k[r])
 *     seg 5 (X_{k+1}[r], u_l) | seg 6 (X_{k+1}[r], dt)   -- each pair once, as (max index, min index).
 * The order is never assumed by a caller: it is reported by pcl_*_structure.
 *
 * Conventions: every function returns 0 (PCL_OK) or a negative pcl_status; the
 * message is available from pcl_last_error.  No exceptions, no abort(), no
 * pointer retained past a call (G0/Gj/x_offs are copied by pcl_create).  The
 * caller owns all Z/delta/vals/mu/rows/cols buffers.  A pcl_ctx is not
 * thread-safe: one context per (host thread x GPU).  Host-pointer entry points
 * are synchronous; *_dev entry points enqueue on the context's stream and return.
 * There is NO CPU fallback: without a gfx950 device pcl_create fails with PCL_EHIP.
 */
#ifndef PICCOLO_HIP_H
#define PICCOLO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcl_ctx pcl_ctx;

typedef enum pcl_status {
    PCL_OK = 0,
    PCL_EINVAL = -1,  /* bad argument / descriptor */
    PCL_ENOMEM = -2,  /* host or device allocation failed */
    PCL_EHIP = -3,    /* HIP runtime error (incl. "no GPU") */
    PCL_ERCCL = -4,   /* RCCL error */
    PCL_ESHAPE = -5,  /* shape outside what the kernels support (d > PCL_MAX_D, ...) */
    PCL_ENOTIMPL = -6, /* valid request the library does not implement (e.g. an odd pade_order) */
    PCL_EINTERNAL = -7 /* a kernel of this context reported a failure on the device (a bounded wait between its waves gave up): the
                          outputs of that launch are incomplete; returned by the next evaluator entry point or pcl_sync, then cleared */
} pcl_status;

#define PCL_MAX_D 32 /* n = 2d <= 64: G(u_k), G^2 and the column tiles stay LDS-resident */

/* batch_mode */
#define PCL_BATCH_MEMBERS 0 /* ONE trajectory buffer; member b owns state columns at x_offs[b]; shared u, dt
                               (SamplingTrajectory: sampling_trajectory.jl:207-237) */
#define PCL_STATE_VECTOR (-1) /* pcl_desc.state_cols: see there */
#define PCL_BATCH_TRAJ 1    /* batch independent trajectory buffers (multistart seeds), Z_b = Z + b*z_dim*N;
                               x_offs[0] is the state offset in each */

typedef struct pcl_desc {
    int32_t struct_size; /* = sizeof(pcl_desc) (ABI check) */
    int32_t d;           /* Hilbert-space dimension (sys.levels); n = 2d, x_dim = 2 d^2 */
    int32_t n_drives;    /* m */
    int32_t N;           /* knot points (traj.N); K = N-1 intervals */
    int32_t z_dim;       /* variables per knot (traj.dim) */
    int32_t u_off;       /* 0-based offset of the drive component inside a knot (traj.components[:u][1]-1) */
    int32_t dt_off;      /* 0-based offset of the timestep component */
    int32_t batch;       /* number of members / seeds (>= 1) */
    int32_t batch_mode;  /* PCL_BATCH_MEMBERS or PCL_BATCH_TRAJ */
    int32_t pade_order;  /* diagonal Pade order p of B^{+-}_p: 2, 4, 6, 8 or 10; 0: the smallest order whose deviation from the reference's
                            exp constraint is below a tolerance -- pcl_set_order_policy, or the first host-pointer call decides */
    int32_t device_id;   /* HIP device ordinal */
    int32_t index_base;  /* 0 (C/Python) or 1 (Julia/MOI) for the emitted structure */
    int32_t per_member_G0; /* 0: one G0 for all members; 1: G0 holds batch matrices (per-member H_drift) */
    int32_t state_cols;  /* columns of the state matrix X (n x state_cols): 0 or d = unitary (Utilde, x_dim = 2 d^2),
                            1 = ket (psitilde = [Re psi; Im psi], x_dim = 2d; KetTrajectory, integrators.jl:58-74);
                            PCL_STATE_VECTOR: the state is a real vector of length d acted on by a general real d x d
                            generator (n = d, odd allowed, d <= 64) -- the compact density vector rhotilde with d = levels^2
                            and the compact Lindbladian generators (DensityTrajectory, integrators.jl:82-95,
                            open_quantum_systems.jl:541-588).  Runs the general-order kernel at every pade_order. */
    int64_t global_dim;  /* traj.global_dim (trailing globals in the variable vector; only shifts nothing, kept for
                            the column count reported by pcl_constraint_dim) */
    const double *G0;      /* n x n column-major (x batch if per_member_G0) : G_drift = iso(-i H_drift) */
    const double *Gj;      /* m matrices n x n column-major                  : G_drives[j] */
    const int32_t *x_offs; /* 0-based state offsets: batch entries (MEMBERS) or 1 entry (TRAJ) */
} pcl_desc;

/* lifecycle ---------------------------------------------------------------- */
int pcl_create(const pcl_desc *desc, pcl_ctx **out);
void pcl_destroy(pcl_ctx *ctx);
/* Message of the last failure on ctx (ctx == NULL: last failure of pcl_create on this thread). */
const char *pcl_last_error(const pcl_ctx *ctx);
const char *pcl_version(void);

/* Order policy.  The reference's constraint is x_{k+1} = exp(dt_k G(u_k)) x_k (docs/src/concepts/index.md:21); the Pade-2q residual deviates from
 * it by kappa_q theta^(2q+1), theta = |dt_k G(u_k)|_2, kappa_q = (q!)^2 / ((2q)! (2q+1)!) (1.6e-5 at order 4, 1.6e-11 at
 * order 8 for BASELINE config 3).  Sets the context's order to the smallest one whose bound at theta = dt_max max_{|u_l| <= u_max[l]} |G_drift + sum_l u_l G_l|_2
 * (the maximum over the box, taken at its 2^m vertices; BASELINE config 3 with |u| <= 0.1, dt <= 0.1: theta = 0.686, order 10 at 1e-10, order 8 at 2e-9)
 * is <= tol and reports it (also: get_option "pade_order").  A context created with pade_order = 0 that never sees this call takes
 * theta = 1.5 x the maximum over the first trajectory a host-pointer entry point is given, tol = 1e-10; its device-pointer entry points
 * return PCL_EINVAL until an order exists. */
int pcl_set_order_policy(pcl_ctx *ctx, double dt_max, const double *u_max /* n_drives */, double tol, int32_t *order_out);
/* The same decision from a trajectory on the HOST (the one the integrator is constructed with -- the reference's constructor,
 * src/control/integrators.jl:35-51, takes it too): theta = 1.5 x max_k |dt_k G(u_k)|_2 over Z_host, tol as set by pcl_set_order_policy
 * (default 1e-10; tol > 0 here overrides).  What a binding calls at construction when the trajectory carries no bounds on u and dt, so that
 * the scalar form B.f, the device-pointer entry points and the host-pointer ones all evaluate ONE order from the first call on.
 * option "order_tol_met" reads 0 (and pcl_last_error carries a note) when even order 10 exceeds the tolerance. */
int pcl_set_order_from_trajectory(pcl_ctx *ctx, const double *Z_host, double tol, int32_t *order_out);
/* The policy itself, without a context or a device: n x n generators (column-major; n = 2 d for unitary and ket problems), n_g0 drifts, m drives.
 * theta_out / order_out / met_out (0: even order 10 exceeds tol) may be NULL. */
int pcl_order_for_bounds(int32_t n, int32_t m, const double *G0, int32_t n_g0, const double *Gj, double dt_max, const double *u_max, double tol,
                         double *theta_out, int32_t *order_out, int32_t *met_out);

/* dimensions --------------------------------------------------------------- */
/* n_rows = batch*x_dim*(N-1); n_cols = z_dim*N*(TRAJ ? batch : 1) + global_dim. Any out pointer may be NULL. */
int pcl_constraint_dim(const pcl_ctx *ctx, int64_t *x_dim, int64_t *n_rows, int64_t *n_cols);
int pcl_jac_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *nnz_per_interval);
int pcl_hess_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *nnz_per_interval);
/* Sparsity structure in value order. In TRAJ mode trajectory b's variables are offset by b*z_dim*N. */
int pcl_jac_structure(const pcl_ctx *ctx, int32_t *rows, int32_t *cols);
int pcl_jac_structure_i64(const pcl_ctx *ctx, int64_t *rows, int64_t *cols);
int pcl_hess_structure(const pcl_ctx *ctx, int32_t *rows, int32_t *cols);
int pcl_hess_structure_i64(const pcl_ctx *ctx, int64_t *rows, int64_t *cols);

/* host-pointer evaluation (synchronous; includes H2D of Z[,mu] and D2H of the results) ---- */
/* Z: z_dim*N doubles (MEMBERS) or batch*z_dim*N (TRAJ). delta: n_rows. vals: jac nnz. */
int pcl_eval(pcl_ctx *ctx, const double *Z, double *delta);
int pcl_jac(pcl_ctx *ctx, const double *Z, double *vals);
int pcl_eval_jac(pcl_ctx *ctx, const double *Z, double *delta, double *vals);
/* mu: n_rows multipliers; vals: hess nnz. (sigma * objective Hessian is the objective's business.) */
int pcl_hess(pcl_ctx *ctx, const double *Z, const double *mu, double *vals);

/* device-pointer evaluation (asynchronous on the context's stream; results stay in HBM) ---- */
/* Launch on the caller's hipStream_t (NULL = HIP's legacy default stream, which is what
 * torch.cuda.current_stream().cuda_stream returns for the default stream). */
int pcl_set_stream(pcl_ctx *ctx, void *hip_stream);
int pcl_reset_stream(pcl_ctx *ctx); /* back to the context's own non-blocking stream (the initial state) */
int pcl_sync(pcl_ctx *ctx);
int pcl_eval_dev(pcl_ctx *ctx, const double *Z_dev, double *delta_dev);
int pcl_eval_jac_dev(pcl_ctx *ctx, const double *Z_dev, double *delta_dev, double *vals_dev);
int pcl_jac_dev(pcl_ctx *ctx, const double *Z_dev, double *vals_dev); /* eval_jacobian alone (integrators.jl:780) */
int pcl_hess_dev(pcl_ctx *ctx, const double *Z_dev, const double *mu_dev, double *vals_dev);

/* Member window: restrict the evaluator entry points (pcl_eval*, pcl_jac*, pcl_hess*, pcl_rollout*, their nnz / structure
 * queries and pcl_constraint_dim's row count) to members / seeds [first, first+count) of the context.  Inputs stay the full
 * buffers (Z of every member; TRAJ mode: of every seed); outputs, multipliers and structure rows are those of the window,
 * numbered from 0.  This is what one member of the reference's Vector{BilinearIntegrator} evaluates
 * (src/control/integrators.jl:134-146: one integrator per ensemble member, rows of each numbered on their own).
 * The objective / merit entry points always cover every member.  Default: all members. */
int pcl_set_member_window(pcl_ctx *ctx, int32_t first, int32_t count);

/* Compact Jacobian: the d diagonal copies of I_d (x) B^{+-} are identical, so per (b,k) only
 * [-B^+ (n*n) | B^- (n*n) | d/du (m*x_dim) | d/ddt (x_dim)] = 2 n^2 + x_dim (m+1) doubles are unique.
 * pcl_eval_jac_compact_dev writes those; pcl_jac_expand_dev replicates them into the full triplet order. */
int pcl_jac_compact_nnz(const pcl_ctx *ctx, int64_t *nnz, int64_t *nnz_per_interval);
int pcl_eval_jac_compact_dev(pcl_ctx *ctx, const double *Z_dev, double *delta_dev, double *compact_dev);
int pcl_jac_expand_dev(pcl_ctx *ctx, const double *compact_dev, double *vals_dev);

/* DerivativeIntegrator / time-consistency rows sharing the context's trajectory layout ------
 *   DerivativeIntegrator(x, dx):  x_{k+1} - x_k - dt_k dx_k = 0      src/control/templates/smooth_pulse_problem.jl:267-275
 *   TimeConsistencyConstraint  :  t_{k+1} - t_k - dt_k      = 0      smooth_pulse_problem.jl:277   (pass dx_off = -1)
 * x_off / dx_off: 0-based component offsets inside a knot, dim: component length.  rows = K*dim (x batch in TRAJ mode),
 * row k*dim + r.  Values per interval: [d/dx_k = -1 (dim) | d/dx_{k+1} = +1 (dim) | d/ddx_k = -dt_k (dim; absent when
 * dx_off < 0) | d/ddt_k = -dx_k[r] (dim)]; order reported by pcl_deriv_structure.  In MEMBERS mode the rows are those of
 * the one shared trajectory (not replicated per member). */
int pcl_deriv_nnz(const pcl_ctx *ctx, int32_t dx_off, int32_t dim, int64_t *n_rows, int64_t *nnz);
int pcl_deriv_structure(const pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, int64_t *rows, int64_t *cols);
int pcl_deriv_eval_jac(pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, const double *Z, double *delta, double *vals);
int pcl_deriv_eval_jac_dev(pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, const double *Z_dev, double *delta_dev,
                           double *vals_dev);

/* objective of the unitary problems -------------------------------------------------------------
 *   UnitaryInfidelityObjective: Q * |1 - F(U_N)|,  F = |tr(U_goal' U_N)|^2 / d^2          src/control/objectives.jl:330-337,347-356
 *   ... with an EmbeddedOperator goal: F = (tr(M'M) + |tr M|^2) / (ns (ns+1)),
 *       M = U_goal[sub,sub]' U_N[sub,sub]                                                   src/control/objectives.jl:339-345
 *   SamplingProblem: sum_i (w_i Q) |1 - F_i| + shared regularisers                          src/control/templates/sampling_problem.jl:381-387
 *   QuadraticRegularizer(name, traj, R) x3 [EXT DirectTrajOpt]                              src/control/templates/smooth_pulse_problem.jl:249-251
 * pcl_set_goal copies the goal's iso-vec (x_dim doubles, operator_to_iso_vec(U_goal)); pcl_set_goal_subspace the iso-vec of
 * the ns x ns block unembed(op) (2 ns^2 doubles) and its ns 0-based subspace indices (op.subspace - 1); the later call wins.
 * pcl_set_weights: per member / seed weights w (batch doubles, NULL = ones).
 * pcl_infidelity_dev writes one value per member / seed (value_dev[batch]) and the gradient w.r.t. that member's terminal
 * state (grad_dev[batch*x_dim], iso-vec order); either output may be NULL.
 * Regularisers: J_r = 1/2 sum_k dt_k^p sum_i R_i Z[k, off+i]^2 with p = dt_power in {0, 1, 2}.  DirectTrajOpt is not vendored
 * with the reference and nothing in the reference pins the value; p = 2 (r_k = dt_k v_k, J += r_k' R r_k / 2) is the form of
 * its QuantumCollocation lineage, p = 0 the plain knot-point form -- the binding chooses.
 * pcl_objective[_dev]: the whole objective and its gradient w.r.t. the variable vector.  MEMBERS mode: value[1] = sum over
 * members + regularisers, grad[z_dim*N].  TRAJ mode: value[batch], grad[batch][z_dim*N] (one NLP per seed).  grad may be NULL. */
int pcl_set_goal(pcl_ctx *ctx, const double *goal_iso_vec);
int pcl_set_goal_subspace(pcl_ctx *ctx, const double *goal_sub_iso_vec, const int32_t *subspace, int32_t ns);
int pcl_set_weights(pcl_ctx *ctx, const double *weights);
int pcl_infidelity_dev(pcl_ctx *ctx, const double *Z_dev, double Q, double *value_dev, double *grad_dev);
int pcl_add_regularizer(pcl_ctx *ctx, int32_t off, int32_t dim, const double *R, int32_t dt_power);
int pcl_clear_regularizers(pcl_ctx *ctx);
/* (one launch -- regulariser rows and terminal infidelities as workgroups of one grid -- with a gradient buffer, the members' states one
 *  contiguous run of a row and no regulariser on a state component; two launches otherwise; the same bits; option "objective_launches") */
int pcl_objective_dev(pcl_ctx *ctx, const double *Z_dev, double Q, double *value_dev, double *grad_dev);
int pcl_objective(pcl_ctx *ctx, const double *Z, double Q, double *value, double *grad);

/* Terminal losses of the other state types, and any loss of the shape  Q w |1 - F(x)|,  F(x) = c'x + sum_r (A_r'x)^2 :
 *   KetInfidelityObjective                        F = |<g|psi>|^2                          src/control/objectives.jl:24-60    (2 rows)
 *   CoherentKetInfidelityObjective                F = |sum_i w_i <g_i|psi_i> / sum w|^2    src/control/objectives.jl:96-200   (2 rows, joint)
 *   DensityMatrix[PureState]InfidelityObjective   F = Re tr(rho rho_goal)                  src/control/objectives.jl:387-435  (linear)
 * x = one member's terminal state (scope 0: a term per member / seed with the weights of pcl_set_weights; A is R x x_dim, row-major, c
 * x_dim doubles or NULL) or the terminal states of ALL members of a MEMBERS context in member order (scope 1: one term; A is
 * R x (batch x_dim)).  The host mirror builds A and c from the goals (piccolo.jl_amd/objectives.py).  Replaces a goal set with
 * pcl_set_goal[_subspace]; pcl_objective[_dev] then evaluates it (three small launches). */
int pcl_set_goal_form(pcl_ctx *ctx, int32_t scope, int32_t R, const double *A, const double *c);

/* Hessian of the objective: sigma * grad^2 f, the part of eval_hessian_lagrangian that pcl_hess does not cover (eval_hessian = true,
 * src/control/templates/spline_pulse_problem.jl:96).  Terminal loss (any goal above, pcl_set_goal and pcl_set_goal_subspace included): per
 * term the lower triangle of -s w Q sigma (2 sum_r A_r A_r'), s = sign(1 - F) -- the Gram triangle is formed once per goal, an evaluation
 * is a scaled copy.  Regularisers: d2/dv_i^2 = dt^p R_i, d2/ddt dv_i = p dt^(p-1) R_i v_i, d2/ddt^2 = p (p - 1) / 2 dt^(p-2) sum_i R_i v_i^2.
 * Each entry once as (max index, min index); the order is reported by pcl_objective_hess_structure (index_base as in pcl_desc). */
int pcl_objective_hess_nnz(const pcl_ctx *ctx, int64_t *nnz);
int pcl_objective_hess_structure(const pcl_ctx *ctx, int64_t *rows, int64_t *cols);
int pcl_objective_hess_dev(pcl_ctx *ctx, const double *Z_dev, double Q, double sigma, double *vals_dev);
int pcl_objective_hess(pcl_ctx *ctx, const double *Z, double Q, double sigma, double *vals);

/* what a sharded ensemble exchanges: out = [phi | g_u (K x m, interval-major) | g_dt (K)] with
 *   phi = sum_b w_b <lam_b, delta_b>  (lam_dev == NULL: lam = delta and phi = 1/2 sum_b w_b |delta_b|^2, the constraint merit),
 *   g = J^T (w lam) restricted to the SHARED variables u_k, dt_k -- the only part of the Lagrangian gradient to which other
 * ranks' members contribute (member states are rank-private).  MEMBERS mode: one set of pcl_merit_grad_len doubles, to be
 * summed over ranks with pcl_reduce_sum_dev; TRAJ mode: `sets` = batch sets (nothing is shared between seeds). */
int pcl_merit_grad_len(const pcl_ctx *ctx, int64_t *len, int64_t *sets);
int pcl_merit_grad_dev(pcl_ctx *ctx, const double *delta_dev, const double *lam_dev, const double *vals_dev, double *out_dev);
/* pcl_eval_jac_dev + pcl_merit_grad_dev in one pass: the fused kernel forms the payload's dot products per state column while the
 * column's vectors are still in LDS (the 1.7 % of the Jacobian values that pcl_merit_grad_dev reads back from HBM are never
 * re-read); delta_dev and vals_dev are written exactly as by pcl_eval_jac_dev.  Launches that do not take fused kernel 3 (other
 * shapes, Pade orders other than 4, a member window) run the two calls one after the other: same outputs, same layout.
 * get_option "last_merit_fused" tells which. */
int pcl_eval_jac_merit_dev(pcl_ctx *ctx, const double *Z_dev, const double *lam_dev, double *delta_dev, double *vals_dev, double *out_dev);
/* pcl_objective_dev + pcl_eval_jac_merit_dev -- everything a rank computes in one step of a sharded ensemble
 * (SamplingProblem: weighted infidelities + regularisers with their gradient, the members' residuals and Jacobian values, the reduce
 * payload) -- in TWO launches instead of four: the fused kernel, then one launch whose workgroups are the regulariser rows, the terminal
 * infidelities and the payload's finish.  Outputs as the two calls write them (value_dev, grad_dev: pcl_objective_dev; the rest:
 * pcl_eval_jac_merit_dev), bit for bit.  Falls back to the two calls where the fused payload does not apply, when grad_dev is NULL, or when a
 * regulariser covers a state component. */
int pcl_eval_jac_merit_objective_dev(pcl_ctx *ctx, const double *Z_dev, const double *lam_dev, double *delta_dev, double *vals_dev, double *out_dev, double Q,
                                     double *value_dev, double *grad_dev);

/* rollout for validation ------------------------------------------------------------------------
 *   unitary_rollout(traj, sys; interpolation = :constant)                       src/quantum/dynamics.jl:631-667
 *   RolloutStates: "a GPU rollout overwrites `states` in place"                 src/quantum/trajectories/ensemble_trajectory.jl:56-71
 * Exact piecewise-constant propagation X_{k+1} = exp(dt_k G(u_k)) X_k from the knot-0 state of every member / trajectory
 * (scaling and squaring, to rounding).  X_out: [batch][N][x_dim] doubles, iso-vec per knot (knot 0 = the input state). */
int pcl_rollout(pcl_ctx *ctx, const double *Z, double *X_out);
int pcl_rollout_dev(pcl_ctx *ctx, const double *Z_dev, double *X_out_dev);

/* multi-GPU: the one exchange of the path ------------------------------------------------------
 * One context per GPU/process.  Rank 0 obtains an id, ships the 128 bytes to the other ranks by any means (MPI,
 * sockets, a file), every rank calls pcl_comm_init; pcl_reduce_sum_dev is an in-place RCCL all-reduce(sum, f64) over
 * xGMI on the context's stream -- used for [merit | d/du | d/ddt] of the shared controls (~5.6 KB: latency-bound).
 * librccl is opened lazily (dlopen); single-GPU users never load it. */
typedef struct pcl_comm_id { char bytes[128]; } pcl_comm_id;
int pcl_comm_get_unique_id(pcl_comm_id *out);
int pcl_comm_init(pcl_ctx *ctx, const pcl_comm_id *id, int32_t rank, int32_t nranks);
int pcl_reduce_sum_dev(pcl_ctx *ctx, double *buf_dev, int64_t n);
int pcl_reduce_sum(pcl_ctx *ctx, double *buf_host, int64_t n); /* same for a host buffer (staged; returns after the sum) */
int pcl_comm_destroy(pcl_ctx *ctx);

/* options --------------------------------------------------------------------
 * A DEPLOYMENT sets none, or only these (everything else is a measurement / test switch, listed with its meaning in OPTIONS.md; every
 * setting gives the same results up to the order of additions inside a kernel family, and inside a family every work split is bitwise
 * equal -- which is what the parity tests use the switches for):
 *   "jit"           1 (default): the pattern-compiled kernels are taken from csrc/prebuilt, from the on-disk cache, or compiled with hiprtc on
 *                   first use; 0: built-in kernel families only
 *   "require_jit"   1: a pattern-compiled kernel that is wanted and cannot be had is PCL_EHIP instead of a (noted, counted) fallback
 *   "host_threads"  threads that expand the compact values into the caller's array in the host-pointer entry points (0 = min(cores / 2, 32);
 *                   -1 = a sweep over the context's first twelve calls), "host_path" (0 auto | 1 full values over PCIe | 2 compact + host expansion)
 *   "v4_ticket"     launches of several trajectories: -1 auto | 0 static split | 1 groups of workgroups + slice tickets
 * and reads: "pade_order" (the order in use), "last_kernel" / "last_hess_kernel" (which kernel family ran), "jit_compiles", "jit_cache_hits",
 * "jit_fallbacks", "n_cu".  Unknown keys return PCL_EINVAL.  Environment: PCL_JIT_CACHE=0, PCL_JIT_CACHE_DIR, PCL_HOST_PATH, PCL_V4_TICKET,
 * PCL_VERBOSE (see OPTIONS.md). */
int pcl_set_option(pcl_ctx *ctx, const char *key, int64_t value);
int pcl_get_option(const pcl_ctx *ctx, const char *key, int64_t *value);

/* Inspection hook (needs no device): the HIP source the library generates and compiles on first use for the
 * pattern-compiled kernels of a system with Hilbert dimension d <= 32 and m <= 6 drives whose generators are exact iso(.)
 * images (G0: n*n column-major, Gj: m such blocks).  *needed = bytes including the terminator; buf may be NULL. */
int pcl_codegen_source(int d, int m, const double *G0, const double *Gj, char *buf, int64_t cap, int64_t *needed);
/* The same for the pattern-compiled FUSED residual + Jacobian kernel (kernel_version 4) at diagonal Pade order 2q, q = 1..5
 * (n_g0 drifts G0[b] span the union pattern: an ensemble's members); PCL_ESHAPE when the drives need more resident
 * coefficients than the kernel keeps. */
int pcl_codegen_source_v4(int d, int m, const double *G0, int n_g0, const double *Gj, int q, int what /* 0 fused residual + Jacobian (+ residual only), 1 Hessian of the Lagrangian (one wave per chain), 5 ... (one wave per group of state columns) */,
                          char *buf, int64_t cap, int64_t *needed);
/* Compile the pattern-compiled module(s) a context of this system would compile on first use and leave the code object under its content
 * hash in out_dir (NULL: <library directory>/prebuilt, which every context looks at before the user's cache and before hiprtc).  Needs
 * libhiprtc, no device.  what: 0 fused residual + Jacobian (+ residual only) at order 2q | 1 general-order Hessian, one workgroup per
 * interval | 2 ... two workgroups per interval | 3 the order-4 Hessian / value-table module (q ignored) | 4 the fused module with the
 * slice-ticket roles (launches of several trajectories) | 5 general-order Hessian, one wave per group of state columns (`auto` at every
 * order but 4).  (6, the resident evaluator's module, in lab builds only: piccolo_hip_lab.h.) */
int pcl_jit_prebuild(int d, int m, const double *G0, int n_g0, const double *Gj, int q, int what, const char *out_dir);

#ifdef __cplusplus
}
#endif
#endif /* PICCOLO_HIP_H */

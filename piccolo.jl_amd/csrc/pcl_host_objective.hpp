// pcl_host_objective.hpp -- part of piccolo_hip.hip (included there, in place): the rows either side of the path -- derivative / time-consistency rows,
// the terminal objectives in their general form with gradient and Hessian, the reduce payload and the fused ensemble step.
#pragma once
// --- DerivativeIntegrator / time-consistency rows ---------------------------
static int deriv_check(const pcl_ctx *ctx, int x_off, int dx_off, int dim) {
    const int zd = ctx->desc.z_dim;
    if (dim < 1 || x_off < 0 || x_off + dim > zd || (dx_off >= 0 && dx_off + dim > zd))
        return fail(ctx, PCL_EINVAL, "derivative rows: components outside the knot (x_off=%d dx_off=%d dim=%d z_dim=%d)", x_off, dx_off, dim, zd);
    return PCL_OK;
}
extern "C" int pcl_deriv_nnz(const pcl_ctx *ctx, int32_t dx_off, int32_t dim, int64_t *rows, int64_t *nnz) {
    if (!ctx) return PCL_EINVAL;
    const long long nb = ctx->desc.batch_mode == PCL_BATCH_TRAJ ? ctx->desc.batch : 1;
    if (rows) *rows = nb * ctx->K * dim;
    if (nnz) *nnz = nb * ctx->K * dim * (dx_off >= 0 ? 4 : 3);
    return PCL_OK;
}
extern "C" int pcl_deriv_structure(const pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, int64_t *rows, int64_t *cols) {
    if (!ctx) return PCL_EINVAL;
    if (!rows || !cols) return fail(ctx, PCL_EINVAL, "pcl_deriv_structure: NULL output");
    TRY(deriv_check(ctx, x_off, dx_off, dim));
    const pcl_desc &D = ctx->desc;
    const long long nb = D.batch_mode == PCL_BATCH_TRAJ ? D.batch : 1, zd = D.z_dim, base = D.index_base;
    const int nseg = dx_off >= 0 ? 4 : 3;
    long long p = 0;
    for (long long b = 0; b < nb; ++b)
        for (long long k = 0; k < ctx->K; ++k) {
            const long long v0 = b * zd * D.N + k * zd + base, r0 = (b * ctx->K + k) * dim + base;
            for (int seg = 0; seg < nseg; ++seg)
                for (long long r = 0; r < dim; ++r, ++p) {
                    rows[p] = r0 + r;
                    cols[p] = seg == 0 ? v0 + x_off + r : seg == 1 ? v0 + zd + x_off + r : (seg == 2 && dx_off >= 0) ? v0 + dx_off + r : v0 + D.dt_off;
                }
        }
    return PCL_OK;
}
extern "C" int pcl_deriv_eval_jac_dev(pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, const double *Z, double *delta,
                                      double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || (!delta && !vals)) return fail(ctx, PCL_EINVAL, "pcl_deriv_eval_jac_dev: NULL pointer");
    TRY(deriv_check(ctx, x_off, dx_off, dim));
    ON_DEVICE(ctx);
    const pcl_desc &D = ctx->desc;
    const long long nb = D.batch_mode == PCL_BATCH_TRAJ ? D.batch : 1;
    const long long total = nb * ctx->K * dim;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pcl_deriv_kernel, dim3(grid), dim3(256), 0, ctx->stream, Z, delta, vals, ctx->K, D.z_dim, x_off, dx_off, dim,
                       D.dt_off, D.batch_mode == PCL_BATCH_TRAJ ? (long long)D.z_dim * D.N : 0LL, total);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
extern "C" int pcl_deriv_eval_jac(pcl_ctx *ctx, int32_t x_off, int32_t dx_off, int32_t dim, const double *Z, double *delta, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || (!delta && !vals)) return fail(ctx, PCL_EINVAL, "pcl_deriv_eval_jac: NULL pointer");
    TRY(deriv_check(ctx, x_off, dx_off, dim));
    ON_DEVICE(ctx);
    int64_t nr = 0, nz = 0;
    pcl_deriv_nnz(ctx, dx_off, dim, &nr, &nz);
    double *dd = nullptr, *dv = nullptr;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    HIP_TRY(ctx, hipMalloc((void **)&dd, (size_t)nr * sizeof(double)));
    if (hipMalloc((void **)&dv, (size_t)nz * sizeof(double)) != hipSuccess) {
        (void)hipFree(dd);
        return fail(ctx, PCL_ENOMEM, "pcl_deriv_eval_jac: device allocation failed");
    }
    int rc = PCL_OK;
    if (hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = PCL_EHIP;
    if (rc == PCL_OK) rc = pcl_deriv_eval_jac_dev(ctx, x_off, dx_off, dim, ctx->dZ, dd, dv);
    if (rc == PCL_OK && delta && hipMemcpyAsync(delta, dd, (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PCL_EHIP;
    if (rc == PCL_OK && vals && hipMemcpyAsync(vals, dv, (size_t)nz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PCL_EHIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == PCL_OK) rc = PCL_EHIP;
    (void)hipFree(dd);
    (void)hipFree(dv);
    if (rc == PCL_EHIP) return fail(ctx, PCL_EHIP, "pcl_deriv_eval_jac: HIP error %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

// --- terminal infidelity objective ----------------------------------------------
static int objective_unitary_only(const pcl_ctx *ctx, const char *who) {
    if (ctx->vec || ctx->cols != ctx->desc.d) return fail(ctx, PCL_ENOTIMPL, "%s: unitary (n x d) states only", who);
    return PCL_OK;
}
// (re)place the general form of the terminal loss; the Gram triangle of the Hessian is formed at the first Hessian call
static int set_form(pcl_ctx *ctx, int scope, int R, const double *A, const double *c, bool user) {
    const long long L = scope ? (long long)ctx->desc.batch * ctx->x_dim : ctx->x_dim;
    for (double **q : {&ctx->dformA, &ctx->dformc, &ctx->dgram, &ctx->dcoef}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    ctx->form_R = ctx->form_L = 0;
    ctx->gram_ready = false;
    ctx->form_user = false;
    if (R < 0 || (R > 0 && !A) || (R == 0 && !c)) return fail(ctx, PCL_EINVAL, "terminal form: need rows or a linear part");
    if (R > 0) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->dformA, (size_t)R * L * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(ctx->dformA, A, (size_t)R * L * sizeof(double), hipMemcpyHostToDevice));
    }
    if (c) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->dformc, (size_t)L * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(ctx->dformc, c, (size_t)L * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMalloc((void **)&ctx->dcoef, (size_t)std::max(ctx->desc.batch, 1) * sizeof(double)));
    ctx->form_R = R;
    ctx->form_L = (int)L;
    ctx->form_scope = scope;
    ctx->form_user = user;
    return PCL_OK;
}
// F = |tr(G'U)|^2 / d^2 = (a'x)^2 + (b'x)^2 with a = iso_vec(G) / d, b = iso_vec(iG) / d   (objectives.jl:330-337)
static int unitary_form(pcl_ctx *ctx, const double *g) {
    const int d = ctx->desc.d, n = ctx->n;
    std::vector<double> A((size_t)2 * ctx->x_dim);
    for (int c = 0; c < d; ++c)
        for (int i = 0; i < d; ++i) {
            const double gr = g[c * n + i], gi = g[c * n + d + i];
            A[c * n + i] = gr / d, A[c * n + d + i] = gi / d;
            A[ctx->x_dim + c * n + i] = -gi / d, A[ctx->x_dim + c * n + d + i] = gr / d;
        }
    return set_form(ctx, 0, 2, A.data(), nullptr, false);
}
// F = (|M|_F^2 + |tr M|^2) / (ns (ns + 1)), M = G_s' U[sub, sub]: a row pair per entry of M and one for the trace   (objectives.jl:339-345)
static int subspace_form(pcl_ctx *ctx, const double *gs, const int32_t *sub, int ns) {
    const int d = ctx->desc.d, n = ctx->n;
    const long long L = ctx->x_dim;
    const int R = 2 * ns * ns + 2;
    std::vector<double> A((size_t)R * L, 0.0);
    const double sc = 1.0 / std::sqrt((double)ns * (ns + 1));
    double *tr_re = A.data() + (size_t)(R - 2) * L, *tr_im = A.data() + (size_t)(R - 1) * L;
    for (int i = 0; i < ns; ++i)
        for (int j = 0; j < ns; ++j) {
            double *re = A.data() + (size_t)(2 * (i * ns + j)) * L, *im = re + L;
            for (int k = 0; k < ns; ++k) {
                const double gr = gs[i * 2 * ns + k] * sc, gi = gs[i * 2 * ns + ns + k] * sc;  // G_s[k, i]
                const long long xr = (long long)sub[j] * n + sub[k], xi = xr + d;               // U[sub_k, sub_j]
                re[xr] += gr, re[xi] += gi;
                im[xi] += gr, im[xr] -= gi;
                if (i == j) tr_re[xr] += gr, tr_re[xi] += gi, tr_im[xi] += gr, tr_im[xr] -= gi;
            }
        }
    return set_form(ctx, 0, R, A.data(), nullptr, false);
}
extern "C" int pcl_set_goal_form(pcl_ctx *ctx, int32_t scope, int32_t R, const double *A, const double *c) {
    if (!ctx) return PCL_EINVAL;
    if (scope != 0 && scope != 1) return fail(ctx, PCL_EINVAL, "pcl_set_goal_form: scope must be 0 (per member) or 1 (joint)");
    if (scope == 1 && ctx->desc.batch_mode != PCL_BATCH_MEMBERS) return fail(ctx, PCL_EINVAL, "pcl_set_goal_form: a joint term needs the members of ONE trajectory buffer");
    if (R > 4096) return fail(ctx, PCL_ESHAPE, "pcl_set_goal_form: at most 4096 rows");
    ON_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->dgoal) (void)hipFree(ctx->dgoal);  // (replaces a unitary goal)
    if (ctx->dsub) (void)hipFree(ctx->dsub);
    ctx->dgoal = nullptr, ctx->dsub = nullptr, ctx->n_sub = 0;
    return set_form(ctx, scope, R, A, c, true);
}
extern "C" int pcl_set_goal(pcl_ctx *ctx, const double *goal_iso_vec) {
    if (!ctx) return PCL_EINVAL;
    if (!goal_iso_vec) return fail(ctx, PCL_EINVAL, "pcl_set_goal: NULL");
    TRY(objective_unitary_only(ctx, "pcl_set_goal"));
    ON_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the context's stream is non-blocking: nothing else orders a launch in flight)
    if (ctx->dgoal) (void)hipFree(ctx->dgoal);
    ctx->dgoal = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->dgoal, (size_t)ctx->x_dim * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dgoal, goal_iso_vec, (size_t)ctx->x_dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_sub = 0;
    return unitary_form(ctx, goal_iso_vec);
}
extern "C" int pcl_set_goal_subspace(pcl_ctx *ctx, const double *goal_sub_iso_vec, const int32_t *subspace, int32_t ns) {
    if (!ctx) return PCL_EINVAL;
    if (!goal_sub_iso_vec || !subspace) return fail(ctx, PCL_EINVAL, "pcl_set_goal_subspace: NULL");
    TRY(objective_unitary_only(ctx, "pcl_set_goal_subspace"));
    if (ns < 1 || ns > ctx->desc.d) return fail(ctx, PCL_EINVAL, "pcl_set_goal_subspace: ns=%d outside 1..d=%d", ns, ctx->desc.d);
    for (int i = 0; i < ns; ++i) {
        if (subspace[i] < 0 || subspace[i] >= ctx->desc.d) return fail(ctx, PCL_EINVAL, "pcl_set_goal_subspace: index %d outside 0..d-1", subspace[i]);
        for (int j = 0; j < i; ++j)
            if (subspace[j] == subspace[i]) return fail(ctx, PCL_EINVAL, "pcl_set_goal_subspace: index %d repeated", subspace[i]);
    }
    ON_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->dgoal) (void)hipFree(ctx->dgoal);
    if (ctx->dsub) (void)hipFree(ctx->dsub);
    ctx->dgoal = nullptr;
    ctx->dsub = nullptr;
    ctx->n_sub = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->dgoal, (size_t)2 * ns * ns * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->dsub, (size_t)ns * sizeof(int)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dgoal, goal_sub_iso_vec, (size_t)2 * ns * ns * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dsub, subspace, (size_t)ns * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_sub = ns;
    return subspace_form(ctx, goal_sub_iso_vec, subspace, ns);
}
extern "C" int pcl_set_weights(pcl_ctx *ctx, const double *w) {
    if (!ctx) return PCL_EINVAL;
    ON_DEVICE(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!w) {
        if (ctx->dweights) (void)hipFree(ctx->dweights);
        ctx->dweights = nullptr;
        return PCL_OK;
    }
    if (!ctx->dweights) HIP_TRY(ctx, hipMalloc((void **)&ctx->dweights, (size_t)ctx->desc.batch * sizeof(double)));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dweights, w, (size_t)ctx->desc.batch * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}
extern "C" int pcl_add_regularizer(pcl_ctx *ctx, int32_t off, int32_t dim, const double *R, int32_t dt_power) {
    if (!ctx) return PCL_EINVAL;
    if (!R || dim < 1 || off < 0 || off + dim > ctx->desc.z_dim) return fail(ctx, PCL_EINVAL, "pcl_add_regularizer: component [%d, %d) outside the knot (z_dim=%d)", off, off + dim, ctx->desc.z_dim);
    if (dt_power < 0 || dt_power > 2) return fail(ctx, PCL_EINVAL, "pcl_add_regularizer: dt_power must be 0, 1 or 2");
    if ((int)ctx->regs.size() >= PCL_MAX_REGS) return fail(ctx, PCL_ESHAPE, "pcl_add_regularizer: at most %d regularisers", PCL_MAX_REGS);
    PclReg r{off, dim, dt_power, (int)ctx->reg_R.size()};
    ctx->regs.push_back(r);
    ctx->reg_R.insert(ctx->reg_R.end(), R, R + dim);
    ctx->regs_dirty = true;
    return PCL_OK;
}
extern "C" int pcl_clear_regularizers(pcl_ctx *ctx) {
    if (!ctx) return PCL_EINVAL;
    ctx->regs.clear();
    ctx->reg_R.clear();
    ctx->regs_dirty = true;
    return PCL_OK;
}
static unsigned infidelity_lds(const pcl_ctx *ctx) { return (unsigned)(6 * (size_t)ctx->n_sub * ctx->n_sub * sizeof(double)); }
extern "C" int pcl_infidelity_dev(pcl_ctx *ctx, const double *Z, double Q, double *value, double *grad) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || (!value && !grad)) return fail(ctx, PCL_EINVAL, "pcl_infidelity_dev: NULL pointer");
    if (!ctx->dgoal) return fail(ctx, PCL_EINVAL, "pcl_infidelity_dev: call pcl_set_goal first");
    TRY(objective_unitary_only(ctx, "pcl_infidelity_dev"));
    ON_DEVICE(ctx);
    const pcl_desc &D = ctx->desc;
    hipLaunchKernelGGL(pcl_infidelity_kernel, dim3((unsigned)D.batch), dim3(256), infidelity_lds(ctx), ctx->stream, Z, ctx->dgoal, ctx->dsub,
                       ctx->n_sub, ctx->dxoffs, ctx->dweights, value, grad, (long long)ctx->x_dim, 0, Q, D.d, D.N, D.z_dim,
                       D.batch_mode == PCL_BATCH_TRAJ ? (long long)D.z_dim * D.N : 0LL, PclObjSum{nullptr, nullptr, nullptr, 0, 0, 0, 0});
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
// Whole objective of the unitary templates: sum_b w_b Q |1 - F_b| + quadratic regularisers, value + full gradient.
// buffers, tickets and the regulariser table of the objective launches (pcl_objective_dev, pcl_eval_jac_merit_objective_dev)
static int objective_prepare(pcl_ctx *ctx) {
    const pcl_desc &D = ctx->desc;
    const bool traj = D.batch_mode == PCL_BATCH_TRAJ;
    const int nbuf = traj ? D.batch : 1;
    if (!ctx->dobj) {  // [member terms | per-knot regulariser values | arrival ticket of the fused final sum]
        HIP_TRY(ctx, hipMalloc((void **)&ctx->dobj, ((size_t)D.batch + (size_t)nbuf * D.N + 1) * sizeof(double)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->dobj, 0, ((size_t)D.batch + (size_t)nbuf * D.N + 1) * sizeof(double), ctx->stream));
    }
    if (ctx->tickets_dirty) {  // (a switch of streams may have left the initial memset pending on the old one)
        HIP_TRY(ctx, hipMemsetAsync(ctx->dobj + D.batch + (size_t)nbuf * D.N, 0, sizeof(double), ctx->stream));
        if (ctx->dmticket) HIP_TRY(ctx, hipMemsetAsync(ctx->dmticket, 0, 64, ctx->stream));
        ctx->tickets_dirty = false;
    }
    if (ctx->regs_dirty) {  // (re)upload the table; rare
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->dregs) (void)hipFree(ctx->dregs);
        if (ctx->dreg_R) (void)hipFree(ctx->dreg_R);
        ctx->dregs = nullptr;
        ctx->dreg_R = nullptr;
        if (!ctx->regs.empty()) {
            HIP_TRY(ctx, hipMalloc((void **)&ctx->dregs, ctx->regs.size() * sizeof(PclReg)));
            HIP_TRY(ctx, hipMalloc((void **)&ctx->dreg_R, ctx->reg_R.size() * sizeof(double)));
            HIP_TRY(ctx, hipMemcpy(ctx->dregs, ctx->regs.data(), ctx->regs.size() * sizeof(PclReg), hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(ctx->dreg_R, ctx->reg_R.data(), ctx->reg_R.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        ctx->regs_dirty = false;
    }
    return PCL_OK;
}
static bool tail_applies(const pcl_ctx *ctx, const double *grad, int &lo_, int &hi_);
static int launch_tail(pcl_ctx *ctx, const double *Z, double Q, double *value, double *grad, int skip_lo, int skip_hi, double *merit_out);
extern "C" int pcl_objective_dev(pcl_ctx *ctx, const double *Z, double Q, double *value, double *grad) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !value) return fail(ctx, PCL_EINVAL, "pcl_objective_dev: NULL pointer");
    if (!ctx->dgoal && !ctx->form_user && ctx->regs.empty()) return fail(ctx, PCL_EINVAL, "pcl_objective_dev: no goal and no regulariser set");
    if (ctx->dgoal) TRY(objective_unitary_only(ctx, "pcl_objective_dev"));
    ON_DEVICE(ctx);
    const pcl_desc &D = ctx->desc;
    const bool traj = D.batch_mode == PCL_BATCH_TRAJ;
    const int nbuf = traj ? D.batch : 1;
    const long long zs = traj ? (long long)D.z_dim * D.N : 0LL;
    if (ctx->form_user) {  // a terminal loss in the general form (kets, coherent kets, densities): regulariser rows, the terms, the sums
        ctx->last_objective_launches = 3;
        TRY(objective_prepare(ctx));
        double *member = ctx->dobj, *regval = ctx->dobj + D.batch;
        hipLaunchKernelGGL(pcl_regularizer_kernel, dim3((unsigned)D.N, (unsigned)nbuf), dim3(256), 0, ctx->stream, Z, (const PclReg *)ctx->dregs,
                           (int)ctx->regs.size(), (const double *)ctx->dreg_R, grad, regval, D.N, D.z_dim, D.dt_off, zs);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemsetAsync(member, 0, (size_t)D.batch * sizeof(double), ctx->stream));
        const PclForm f{ctx->dformA, ctx->dformc, ctx->form_R, ctx->form_L, ctx->form_scope};
        hipLaunchKernelGGL(pcl_form_kernel, dim3(ctx->form_scope ? 1u : (unsigned)D.batch), dim3(256), (size_t)std::max(ctx->form_R, 1) * sizeof(double), ctx->stream, Z, f,
                           (const int *)ctx->dxoffs, (const double *)ctx->dweights, Q, 1.0, (int)ctx->x_dim, D.N, D.z_dim, zs, (long long)D.z_dim * D.N, member, grad, (double *)nullptr);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(pcl_objective_sum_kernel, dim3(traj ? (unsigned)D.batch : 1u), dim3(64), 0, ctx->stream, (const double *)member, (const double *)regval, value,
                           D.batch, D.N, traj ? 1 : 0);
        HIP_TRY(ctx, hipGetLastError());
        return PCL_OK;
    }
    {  // ONE launch where it applies (regulariser rows and terminal infidelities as workgroups of one grid: 18 -> 10 us); the same bits
        int lo = 0, hi = 0;
        if (tail_applies(ctx, grad, lo, hi)) {
            ctx->last_objective_launches = 1;
            return launch_tail(ctx, Z, Q, value, grad, lo, hi, nullptr);
        }
    }
    ctx->last_objective_launches = 2;
    TRY(objective_prepare(ctx));
    double *member = ctx->dobj, *regval = ctx->dobj + D.batch;
    // Two launches: the regulariser kernel writes every knot's whole gradient row (zeros where no term applies) and the per-knot
    // values; the infidelity kernel adds the terminal-state blocks and its last-arriving workgroup forms the final sum(s).
    hipLaunchKernelGGL(pcl_regularizer_kernel, dim3((unsigned)D.N, (unsigned)nbuf), dim3(256), 0, ctx->stream, Z, (const PclReg *)ctx->dregs,
                       (int)ctx->regs.size(), (const double *)ctx->dreg_R, grad, regval, D.N, D.z_dim, D.dt_off, zs);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->dgoal) {
        hipLaunchKernelGGL(pcl_infidelity_kernel, dim3((unsigned)D.batch), dim3(256), infidelity_lds(ctx), ctx->stream, Z, ctx->dgoal,
                           ctx->dsub, ctx->n_sub, ctx->dxoffs, ctx->dweights, member, grad, traj ? (long long)D.z_dim * D.N : 0LL, 1, Q, D.d,
                           D.N, D.z_dim, zs,
                           PclObjSum{value, regval, reinterpret_cast<unsigned int *>(ctx->dobj + D.batch + (size_t)nbuf * D.N), D.batch, D.N, traj ? 1 : 0, D.batch});
        HIP_TRY(ctx, hipGetLastError());
        return PCL_OK;
    }
    HIP_TRY(ctx, hipMemsetAsync(member, 0, (size_t)D.batch * sizeof(double), ctx->stream));  // regularisers only
    hipLaunchKernelGGL(pcl_objective_sum_kernel, dim3(traj ? (unsigned)D.batch : 1u), dim3(64), 0, ctx->stream, (const double *)member, (const double *)regval, value,
                       D.batch, D.N, traj ? 1 : 0);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
extern "C" int pcl_objective(pcl_ctx *ctx, const double *Z, double Q, double *value, double *grad) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !value) return fail(ctx, PCL_EINVAL, "pcl_objective: NULL pointer");
    ON_DEVICE(ctx);
    const int nval = ctx->desc.batch_mode == PCL_BATCH_TRAJ ? ctx->desc.batch : 1;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    TRY(ensure(ctx, &ctx->dgrad, z_len(ctx)));
    TRY(ensure(ctx, &ctx->dval, nval));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TRY(pcl_objective_dev(ctx, ctx->dZ, Q, ctx->dval, ctx->dgrad));
    HIP_TRY(ctx, hipMemcpyAsync(value, ctx->dval, (size_t)nval * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (grad) HIP_TRY(ctx, hipMemcpyAsync(grad, ctx->dgrad, z_len(ctx) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}
// [phi | J^T lam on the shared controls and time steps]: the payload of the one collective (pcl_reduce_sum_dev)
extern "C" int pcl_merit_grad_dev(pcl_ctx *ctx, const double *delta, const double *lam, const double *vals, double *out) {
    if (!ctx) return PCL_EINVAL;
    if (!delta || !vals || !out) return fail(ctx, PCL_EINVAL, "pcl_merit_grad_dev: NULL pointer");
    ON_DEVICE(ctx);
    const pcl_desc &D = ctx->desc;
    const bool traj = D.batch_mode == PCL_BATCH_TRAJ;
    const int sets = traj ? D.batch : 1;
    const int m = D.n_drives;
    if (!ctx->dphik) HIP_TRY(ctx, hipMalloc((void **)&ctx->dphik, ((size_t)D.batch * ctx->K * (m + 2) + (size_t)sets * ctx->K) * sizeof(double)));
    double *part = ctx->dphik, *phik = ctx->dphik + (size_t)D.batch * ctx->K * (m + 2);
    hipLaunchKernelGGL(pcl_merit_part_kernel, dim3((unsigned)ctx->K, (unsigned)D.batch), dim3(512), 0, ctx->stream, delta, lam, vals, part,
                       ctx->K, ctx->cols, ctx->n, m, jac_per_full(ctx), 2LL * ctx->cols * ctx->n * ctx->n);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(pcl_merit_sum_kernel, dim3((unsigned)sets), dim3(1024), 0, ctx->stream, (const double *)part,
                       (const double *)ctx->dweights, out, phik, D.batch, ctx->K, m, traj ? 1 : 0);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
// fused residual + Jacobian + reduce payload: one pass over the state columns (the tails are not read back from HBM)
extern "C" int pcl_eval_jac_merit_dev(pcl_ctx *ctx, const double *Z, const double *lam, double *delta, double *vals, double *out) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !delta || !vals || !out) return fail(ctx, PCL_EINVAL, "pcl_eval_jac_merit_dev: NULL pointer");
    ctx->merit_want = 1;
    ctx->merit_fused = 0;
    ctx->merit_lam = lam;
    const int rc = launch_fused(ctx, Z, delta, vals, false);
    ctx->merit_want = 0;
    ctx->merit_lam = nullptr;
    if (rc != PCL_OK) return rc;
    if (!ctx->merit_fused) return pcl_merit_grad_dev(ctx, delta, lam, vals, out);  // other kernels / member windows: the separate payload kernels
    ON_DEVICE(ctx);
    const pcl_desc &D = ctx->desc;
    const bool traj = D.batch_mode == PCL_BATCH_TRAJ;
    const int sets = traj ? D.batch : 1;
    const int m = D.n_drives;
    if (!ctx->dphik) HIP_TRY(ctx, hipMalloc((void **)&ctx->dphik, ((size_t)D.batch * ctx->K * (m + 2) + (size_t)sets * ctx->K) * sizeof(double)));
    double *phik = ctx->dphik + (size_t)D.batch * ctx->K * (m + 2);
    if (!ctx->dmticket) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->dmticket, 64));
        HIP_TRY(ctx, hipMemsetAsync(ctx->dmticket, 0, 64, ctx->stream));
    } else if (ctx->tickets_dirty) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->dmticket, 0, 64, ctx->stream));
        if (ctx->dobj) {
            const int nbuf_ = traj ? D.batch : 1;
            HIP_TRY(ctx, hipMemsetAsync(ctx->dobj + D.batch + (size_t)nbuf_ * D.N, 0, sizeof(double), ctx->stream));
        }
        ctx->tickets_dirty = false;
    }
    // ONE launch: a workgroup per interval adds the columns, then the members (weights, member order); the workgroup that
    // arrives last adds phi over the intervals
    hipLaunchKernelGGL(pcl_merit_finish_kernel, dim3((unsigned)ctx->K), dim3(256), (size_t)D.batch * (m + 2) * sizeof(double), ctx->stream,
                       (const double *)ctx->dmcols, (const double *)ctx->dweights, out, phik, ctx->dmticket, D.batch, ctx->K, ctx->cols, m, traj ? 1 : 0);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
// The one-launch tail (pcl_ens_tail_kernel: regulariser rows + terminal infidelities [+ the payload's finish]) applies with a gradient
// buffer, when the members' states are ONE contiguous run of a gradient row (the regulariser workgroup of the last knot leaves that run to
// the infidelity workgroups of the same launch) that no regulariser covers (its terminal-knot term and the infidelity's would meet in one
// entry: the launches then have to stay in order).  [lo, hi) = that run.
// --- Hessian of the objective (what eval_hessian_lagrangian adds to the constraints' term: sigma * grad^2 f) -----------------------------
// values: [terminal blocks: per term the lower triangle (i, j <= i) of its L x L block] [per buffer, knot, regulariser:
// d2/dv_i^2 (dim) | d2/ddt dv_i (dim, dt_power >= 1) | d2/ddt^2 (dt_power 2)]; the structure says where each value belongs.
static long long obj_hess_terms(const pcl_ctx *ctx) { return (ctx->dformA || ctx->dformc) ? (ctx->form_scope ? 1 : ctx->desc.batch) : 0; }
static long long obj_hess_tri(const pcl_ctx *ctx) { return ctx->form_R > 0 ? (long long)ctx->form_L * (ctx->form_L + 1) / 2 : 0; }  // (a linear form has no second derivative)
static long long obj_hess_per_knot(const pcl_ctx *ctx) {
    long long n = 0;
    for (const PclReg &r : ctx->regs) n += (long long)r.dim * (r.pw >= 1 ? 2 : 1) + (r.pw == 2 ? 1 : 0);
    return n;
}
extern "C" int pcl_objective_hess_nnz(const pcl_ctx *ctx, int64_t *nnz) {
    if (!ctx || !nnz) return PCL_EINVAL;
    const int nbuf = ctx->desc.batch_mode == PCL_BATCH_TRAJ ? ctx->desc.batch : 1;
    *nnz = obj_hess_terms(ctx) * obj_hess_tri(ctx) + (long long)nbuf * ctx->desc.N * obj_hess_per_knot(ctx);
    return PCL_OK;
}
extern "C" int pcl_objective_hess_structure(const pcl_ctx *ctx, int64_t *rows, int64_t *cols) {
    if (!ctx || !rows || !cols) return PCL_EINVAL;
    const pcl_desc &D = ctx->desc;
    const bool traj = D.batch_mode == PCL_BATCH_TRAJ;
    const int nbuf = traj ? D.batch : 1;
    const long long base = D.index_base, zn = (long long)D.z_dim * D.N;
    long long e = 0;
    auto put = [&](long long a, long long b) {
        rows[e] = std::max(a, b) + base;
        cols[e] = std::min(a, b) + base;
        ++e;
    };
    auto var = [&](long long t, long long i) {  // element i of term t's argument
        const long long mem = ctx->form_scope ? i / ctx->x_dim : t, r = ctx->form_scope ? i - mem * ctx->x_dim : i;
        return (traj ? mem * zn + ctx->x_offs[0] : (long long)ctx->x_offs[mem]) + (long long)(D.N - 1) * D.z_dim + r;
    };
    if (obj_hess_tri(ctx))
        for (long long t = 0; t < obj_hess_terms(ctx); ++t)
            for (long long i = 0; i < ctx->form_L; ++i)
                for (long long j = 0; j <= i; ++j) put(var(t, i), var(t, j));
    for (int b = 0; b < nbuf; ++b)
        for (int k = 0; k < D.N; ++k) {
            const long long z0 = (long long)b * zn + (long long)k * D.z_dim;
            for (const PclReg &r : ctx->regs) {
                for (int i = 0; i < r.dim; ++i) put(z0 + r.off + i, z0 + r.off + i);
                if (r.pw >= 1)
                    for (int i = 0; i < r.dim; ++i) put(z0 + D.dt_off, z0 + r.off + i);
                if (r.pw == 2) put(z0 + D.dt_off, z0 + D.dt_off);
            }
        }
    return PCL_OK;
}
extern "C" int pcl_objective_hess_dev(pcl_ctx *ctx, const double *Z, double Q, double sigma, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !vals) return fail(ctx, PCL_EINVAL, "pcl_objective_hess_dev: NULL pointer");
    ON_DEVICE(ctx);
    const pcl_desc &D = ctx->desc;
    const bool traj = D.batch_mode == PCL_BATCH_TRAJ;
    const int nbuf = traj ? D.batch : 1;
    const long long zs = traj ? (long long)D.z_dim * D.N : 0LL;
    const long long nT = obj_hess_tri(ctx), nterm = obj_hess_terms(ctx);
    if (nT) {
        const PclForm f{ctx->dformA, ctx->dformc, ctx->form_R, ctx->form_L, ctx->form_scope};
        if (!ctx->gram_ready) {  // T = 2 sum_r A_r A_r', once per goal
            if (!ctx->dgram) HIP_TRY(ctx, hipMalloc((void **)&ctx->dgram, (size_t)nT * sizeof(double)));
            hipLaunchKernelGGL(pcl_gram_kernel, dim3((unsigned)std::min<long long>((nT + 255) / 256, 4096)), dim3(256), 0, ctx->stream, f, ctx->dgram);
            HIP_TRY(ctx, hipGetLastError());
            ctx->gram_ready = true;
        }
        hipLaunchKernelGGL(pcl_form_kernel, dim3((unsigned)nterm), dim3(256), (size_t)std::max(ctx->form_R, 1) * sizeof(double), ctx->stream, Z, f, (const int *)ctx->dxoffs,
                           (const double *)ctx->dweights, Q, sigma, (int)ctx->x_dim, D.N, D.z_dim, zs, (long long)D.z_dim * D.N, (double *)nullptr, (double *)nullptr, ctx->dcoef);
        HIP_TRY(ctx, hipGetLastError());
        hipLaunchKernelGGL(pcl_scale_kernel, dim3((unsigned)std::min<long long>((nT * nterm + 255) / 256, 8192)), dim3(256), 0, ctx->stream, (const double *)ctx->dgram,
                           (const double *)ctx->dcoef, nT, (int)nterm, vals);
        HIP_TRY(ctx, hipGetLastError());
    }
    const long long pk = obj_hess_per_knot(ctx);
    if (pk) {
        TRY(objective_prepare(ctx));
        hipLaunchKernelGGL(pcl_reg_hess_kernel, dim3((unsigned)D.N, (unsigned)nbuf), dim3(256), 0, ctx->stream, Z, (const PclReg *)ctx->dregs, (int)ctx->regs.size(),
                           (const double *)ctx->dreg_R, sigma, D.N, D.z_dim, D.dt_off, zs, pk, vals + nT * nterm);
        HIP_TRY(ctx, hipGetLastError());
    }
    return PCL_OK;
}
extern "C" int pcl_objective_hess(pcl_ctx *ctx, const double *Z, double Q, double sigma, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !vals) return fail(ctx, PCL_EINVAL, "pcl_objective_hess: NULL pointer");
    ON_DEVICE(ctx);
    int64_t nnz = 0;
    TRY(pcl_objective_hess_nnz(ctx, &nnz));
    if (nnz == 0) return PCL_OK;
    TRY(ensure(ctx, &ctx->dZ, z_len(ctx)));
    double *dv = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&dv, (size_t)nnz * sizeof(double)));
    int rc = PCL_OK;
    if (hipMemcpyAsync(ctx->dZ, Z, z_len(ctx) * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = PCL_EHIP;
    if (rc == PCL_OK) rc = pcl_objective_hess_dev(ctx, ctx->dZ, Q, sigma, dv);
    if (rc == PCL_OK && hipMemcpyAsync(vals, dv, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PCL_EHIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == PCL_OK) rc = PCL_EHIP;
    (void)hipFree(dv);
    if (rc == PCL_EHIP) return fail(ctx, PCL_EHIP, "pcl_objective_hess: HIP error %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

static bool tail_applies(const pcl_ctx *ctx, const double *grad, int &lo_, int &hi_) {
    if (!grad || !ctx->dgoal || ctx->opt_objective_launches == 2) return false;
    const int nx = ctx->desc.batch_mode == PCL_BATCH_TRAJ ? 1 : ctx->desc.batch;
    // The members' states must TILE one run [lo, hi) of the knot's row: sorted, every neighbour exactly x_dim further (distinct offsets and
    // a matching extent are not enough: x_dim = 4 with offsets {0, 2, 8} has both and leaves a gap behind two overlapping states)
    std::vector<long long> xo(ctx->x_offs.begin(), ctx->x_offs.begin() + nx);
    std::sort(xo.begin(), xo.end());
    for (int b = 1; b < nx; ++b)
        if (xo[b] - xo[b - 1] != (long long)ctx->x_dim) return false;
    const long long lo = xo[0], hi = xo[nx - 1] + ctx->x_dim;
    for (const PclReg &r : ctx->regs)
        if (r.off < hi && lo < r.off + r.dim) return false;
    lo_ = (int)lo;
    hi_ = (int)hi;
    return true;
}
static int launch_tail(pcl_ctx *ctx, const double *Z, double Q, double *value, double *grad, int skip_lo, int skip_hi, double *merit_out) {
    const pcl_desc &D = ctx->desc;
    const bool traj = D.batch_mode == PCL_BATCH_TRAJ;
    const int sets = traj ? D.batch : 1, nbuf = sets;
    const int m = D.n_drives;
    const long long zs = traj ? (long long)D.z_dim * D.N : 0LL;
    if (merit_out) {
        if (!ctx->dphik) HIP_TRY(ctx, hipMalloc((void **)&ctx->dphik, ((size_t)D.batch * ctx->K * (m + 2) + (size_t)sets * ctx->K) * sizeof(double)));
        if (!ctx->dmticket) {
            HIP_TRY(ctx, hipMalloc((void **)&ctx->dmticket, 64));
            HIP_TRY(ctx, hipMemsetAsync(ctx->dmticket, 0, 64, ctx->stream));
        }
    }
    TRY(objective_prepare(ctx));  // (also re-zeroes both tickets after a switch of streams)
    PclTailArgs a;
    a.Z = Z;
    a.regs = (const PclReg *)ctx->dregs;
    a.n_regs = (int)ctx->regs.size();
    a.Rv = ctx->dreg_R;
    a.grad = grad;
    a.regval = ctx->dobj + D.batch;
    a.N = D.N;
    a.z_dim = D.z_dim;
    a.dt_off = D.dt_off;
    a.nbuf = nbuf;
    a.z_batch_stride = zs;
    a.goal = ctx->dgoal;
    a.sub = ctx->dsub;
    a.ns = ctx->n_sub;
    a.x_offs = ctx->dxoffs;
    a.weights = ctx->dweights;
    a.member = ctx->dobj;
    a.grad_stride = traj ? (long long)D.z_dim * D.N : 0LL;
    a.Q = Q;
    a.d = D.d;
    a.batch = D.batch;
    a.fin = PclObjSum{value, ctx->dobj + D.batch, reinterpret_cast<unsigned int *>(ctx->dobj + D.batch + (size_t)nbuf * D.N), D.batch, D.N, traj ? 1 : 0,
                      D.batch + nbuf * D.N};
    a.pcol = ctx->dmcols;
    a.out = merit_out;
    a.phik = merit_out ? ctx->dphik + (size_t)D.batch * ctx->K * (m + 2) : nullptr;
    a.mticket = ctx->dmticket;
    a.K = merit_out ? ctx->K : 0;
    a.cols = ctx->cols;
    a.m = m;
    a.traj_mode = traj ? 1 : 0;
    a.skip_lo = skip_lo;
    a.skip_hi = skip_hi;
    const size_t lds = std::max((size_t)infidelity_lds(ctx), merit_out ? (size_t)D.batch * (m + 2) * sizeof(double) : (size_t)0);
    hipLaunchKernelGGL(pcl_ens_tail_kernel, dim3((unsigned)(nbuf * D.N + D.batch + a.K)), dim3(256), lds, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return PCL_OK;
}
// pcl_objective_dev + pcl_eval_jac_merit_dev as TWO launches instead of four: the fused kernel, then ONE launch whose workgroups are the
// regulariser rows, the terminal infidelities and the payload's finish (pcl_ens_tail_kernel); the same bits as the separate calls.
extern "C" int pcl_eval_jac_merit_objective_dev(pcl_ctx *ctx, const double *Z, const double *lam, double *delta, double *vals, double *out, double Q,
                                                double *value, double *grad) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !delta || !vals || !out || !value) return fail(ctx, PCL_EINVAL, "pcl_eval_jac_merit_objective_dev: NULL pointer");
    if (!ctx->dgoal) return fail(ctx, PCL_EINVAL, "pcl_eval_jac_merit_objective_dev: no goal set");
    TRY(objective_unitary_only(ctx, "pcl_eval_jac_merit_objective_dev"));
    int skip_lo = 0, skip_hi = 0;
    ctx->last_step_launches = 4;
    if (!tail_applies(ctx, grad, skip_lo, skip_hi)) {
        TRY(pcl_objective_dev(ctx, Z, Q, value, grad));
        ctx->last_step_launches = 2 + ctx->last_objective_launches;
        return pcl_eval_jac_merit_dev(ctx, Z, lam, delta, vals, out);
    }
    ctx->merit_want = 1;
    ctx->merit_fused = 0;
    ctx->merit_lam = lam;
    const int rc = launch_fused(ctx, Z, delta, vals, false);
    ctx->merit_want = 0;
    ctx->merit_lam = nullptr;
    if (rc != PCL_OK) return rc;
    if (!ctx->merit_fused) {  // other kernels / member windows: the separate calls
        TRY(pcl_merit_grad_dev(ctx, delta, lam, vals, out));
        return pcl_objective_dev(ctx, Z, Q, value, grad);
    }
    ON_DEVICE(ctx);
    TRY(launch_tail(ctx, Z, Q, value, grad, skip_lo, skip_hi, out));
    ctx->last_step_launches = 2;
    return PCL_OK;
}
extern "C" int pcl_merit_grad_len(const pcl_ctx *ctx, int64_t *len, int64_t *sets) {
    if (!ctx) return PCL_EINVAL;
    if (len) *len = 1 + (int64_t)ctx->K * ctx->desc.n_drives + ctx->K;
    if (sets) *sets = ctx->desc.batch_mode == PCL_BATCH_TRAJ ? ctx->desc.batch : 1;
    return PCL_OK;
}

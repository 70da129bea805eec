// pcl_host_options.hpp -- part of piccolo_hip.hip (included there, in place): pcl_set_option / pcl_get_option / pcl_debug_timing (the switches are
// listed in OPTIONS.md).
#pragma once
// --- options ---------------------------------------------------------------------------------
extern "C" int pcl_set_option(pcl_ctx *ctx, const char *key, int64_t v) {
    if (!ctx || !key) return PCL_EINVAL;
    if (!strcmp(key, "cols_per_slice")) {
        if (v < 0) return fail(ctx, PCL_EINVAL, "cols_per_slice must be >= 0");
        ctx->opt_cols_per_slice = v;
    } else if (!strcmp(key, "use_mfma"))
        ctx->opt_use_mfma = v != 0;
    else if (!strcmp(key, "nt_stores")) {  // -1 auto (by launch size) | 0 plain | 1 nontemporal | 2 write-through
        if (v < -1 || v > 3) return fail(ctx, PCL_EINVAL, "nt_stores must be -1, 0, 1, 2 or 3");
        ctx->opt_nt = v;
    }
    else if (!strcmp(key, "grid"))
        ctx->opt_grid = v;
#ifdef PCL_PROFILE
    else if (!strcmp(key, "profile_flags"))  // profiling experiments (results may be WRONG); not present in the shipped library
        ctx->opt_prof = v;
    else if (!strcmp(key, "v4_variant")) {  // timing variants of kernel 4's generated product (WRONG results); recompiles
        ctx->opt_v4_variant = v;
        ctx->v4_f = ctx->v4_ft = ctx->v4_feval = ctx->v4_fevalc = ctx->v4_fhess = ctx->v4_fhess2 = ctx->v4_fhessc = nullptr;
    }
#endif
    else if (!strcmp(key, "host_threads"))  // host-pointer entry points: threads expanding the compact values (0 = auto)
    {
        ctx->opt_host_threads = v < 0 ? -1 : v;  // (0 default count | n | -1 sweep over the context's first twelve calls)
        ctx->host_threads_tuned = 0;
        ctx->host_tune_calls = 0;
        for (double &t : ctx->host_tune_t) t = 1e300;
    }
    else if (!strcmp(key, "host_path"))  // 0 auto | 1 full values over PCIe | 2 compact values + host expansion
        ctx->opt_host_path = v < 0 || v > 2 ? 0 : v;
    else if (!strcmp(key, "host_chunks"))  // interval chunks of the compact D2H copy that overlap with the expansion (1..8)
        ctx->opt_host_chunks = v < 1 ? 1 : (v > 8 ? 8 : v);
    else if (!strcmp(key, "specialize"))  // 0: always the run-time-shape kernel instances
        ctx->opt_specialize = v != 0;
    else if (!strcmp(key, "jit"))  // 1 (default): compile the context's shape on first use when no static instance matches
        ctx->opt_jit = v != 0;
    else if (!strcmp(key, "require_jit"))  // 1: a pattern-compiled kernel that is wanted and cannot be had is an error (PCL_EHIP), not a fallback
        ctx->opt_require_jit = v != 0;
    else if (!strcmp(key, "general_threads"))  // general-order kernel: 256 or 512 (default) threads per workgroup
        ctx->opt_general_threads = v == 256 ? 256 : 512;
    else if (!strcmp(key, "general_kernel_version"))  // general-order residual+Jacobian: 0 auto | 1 reference formulation | 2 lock-step kernel (error where it does not fit)
        ctx->opt_general_version = v < 0 || v > 2 ? 0 : v;
    else if (!strcmp(key, "general_slices"))  // lock-step kernel: slices of state columns per interval (0 auto)
        ctx->opt_general_slices = v < 0 ? 0 : v;
    else if (!strcmp(key, "general_pade_kernel"))  // 1: the general-order kernel also for pade_order 4
        ctx->opt_general = v != 0;
    else if (!strcmp(key, "stream_workgroups"))  // kernel 3, contiguous: > 0 = role split with this many stream-role workgroups
        ctx->opt_stream_wg = v;
    else if (!strcmp(key, "contiguous"))  // kernel 3: 1 = equal contiguous column ranges per workgroup (default), 0 = round-robin slices
        ctx->opt_contig = v < 0 ? -1 : (v != 0);
    else if (!strcmp(key, "objective_launches")) {  // 0 auto (one launch where it applies) | 2 always regulariser + infidelity launches
        if (v != 0 && v != 2) return fail(ctx, PCL_EINVAL, "objective_launches must be 0 or 2");
        ctx->opt_objective_launches = v;
    } else if (!strcmp(key, "v4_flags"))  // kernel 4 A/B switches: 1 no raised priority for the P wave | 2 tails only behind the item's last block
                                       // | 4 no cooperative first item | 8 LDS tiles NaN at kernel start (tests) | 16 the first item's chains do not wait for the cooperative products | 32 no balanced split of the middle column's two blocks between two slices
        ctx->opt_v4_flags = v;
    else if (!strcmp(key, "v4_ticket"))  // kernel 4: work items by ticket (-1 auto: full-value launches of several intervals per CU | 0 static split | 1)
        ctx->opt_v4_ticket = v < 0 ? -1 : (v != 0);
    else if (!strcmp(key, "v4_ticket_cols"))  // ... state columns per slice ticket (0 auto: 4 since round 5; 3 in round 4)
        ctx->opt_v4_ticket_cols = v < 0 ? 0 : v;
    else if (!strcmp(key, "v4_ticket_ahead"))  // ... slices taken ahead of the one being stored (0 | 1)
        ctx->opt_v4_ticket_ahead = v < 0 || v > 2 ? 0 : v;
    else if (!strcmp(key, "v4_group"))  // ... workgroups per group (0 auto: 8; must divide the grid)
        ctx->opt_v4_group = v < 0 ? 0 : v;
    else if (!strcmp(key, "v4_power_tiles"))  // kernel 4: LDS tiles the powers of G rotate through (0 auto: q - 1 for launches of several items per workgroup, else q)
        ctx->opt_v4_np = v < 0 ? 0 : v;
    else if (!strcmp(key, "v4_tail_mode")) {  // kernel 4: 0 writer wave, plain stores | 1 nontemporal | 2 write-through | 3 the stream waves store the tails
        if (v < 0 || v > 3) return fail(ctx, PCL_EINVAL, "v4_tail_mode must be 0 .. 3");
        ctx->opt_v4_tail_mode = v;
    }
    else if (!strcmp(key, "v4_tune"))  // v4_ticket auto: 1 decide static split / slice tickets per values array by timing both on it | 0 always tickets
        ctx->opt_v4_tune = v != 0;
    else if (!strcmp(key, "host_store_bytes"))  // host expansion: bytes per streaming store (0 the widest the host has | 16 | 32 | 64)
        ctx->opt_host_store_bytes = (v == 16 || v == 32 || v == 64) ? v : 0;
#ifdef PCL_LAB
    else if (!strcmp(key, "resident_idle_us"))  // resident evaluator: the kernel leaves after this long without a request (10 .. 2 000 000)
        ctx->opt_resident_idle_us = v < 10 ? 10 : (v > 2000000 ? 2000000 : v);
#endif
    else if (!strcmp(key, "hess_xcd"))  // column-group Hessian kernel: the waves of an interval take blockIdx values equal mod n = one XCD (-1 auto: 8 | 0, 1: blockIdx order)
        ctx->opt_hess_xcd = v < 0 ? -1 : v;
    else if (!strcmp(key, "hess_rpre"))  // column-group Hessian kernel: the chain of the R_a in R-chain waves at the head of the launch (-1 auto | 0 | 1)
        ctx->opt_hess_rpre = v < 0 ? -1 : (v ? 1 : 0);
    else if (!strcmp(key, "hess_pair"))  // column-group Hessian kernel: a chain wave and a contribution wave per column group (-1 auto: launches of at most n_cu / 2 intervals | 0 | 1)
        ctx->opt_hess_pair = v < 0 ? -1 : (v != 0);
    else if (!strcmp(key, "hess_split"))  // general-order pattern-compiled Hessian kernel: two workgroups per interval (-1 auto by launch size | 0 | 1)
        ctx->opt_hess_split = v < 0 ? -1 : (v != 0);
    else if (!strcmp(key, "eval_coop"))  // pattern-compiled residual kernel: four waves per interval (-1 auto by launch size | 0 | 1)
        ctx->opt_eval_coop = v < 0 ? -1 : (v != 0);
    else if (!strcmp(key, "eval_kernel")) {  // residual only: 0 auto, 1 matrix-core kernel, 2 pattern-compiled kernel
        if (v < 0 || v > 3) return fail(ctx, PCL_EINVAL, "eval_kernel must be 0 .. 3");
        ctx->opt_eval_kernel = v;
    }
    else if (!strcmp(key, "hess_kernel")) {  // 0: auto, 1: one workgroup per interval, 2: persistent wave-synchronous kernel
        if ((v < 0 || v > 4) && v != 7 && v != 8) return fail(ctx, PCL_EINVAL, "hess_kernel must be 0 .. 4, 7 or 8");
        ctx->opt_hess_kernel = v;
    }
    else if (!strcmp(key, "debug_timing")) {  // profiling aid: cycle stamps of workgroup 0 (pcl_debug_timing reads them)
#ifndef PCL_PROFILE
        if (v) return fail(ctx, PCL_ENOTIMPL, "debug_timing needs a library built with -DPCL_PROFILE (the shipped kernels carry no stamps)");
#endif
        if (v && !ctx->ddbg) {
            HIP_TRY(ctx, hipMalloc((void **)&ctx->ddbg, PCL_DBG_WORDS * sizeof(long long)));
            HIP_TRY(ctx, hipMemset(ctx->ddbg, 0, PCL_DBG_WORDS * sizeof(long long)));
        } else if (!v && ctx->ddbg) {
            (void)hipFree(ctx->ddbg);
            ctx->ddbg = nullptr;
        }
    }
    else if (!strcmp(key, "kernel_version")) {
        if (v < 0 || v > 5) return fail(ctx, PCL_EINVAL, "kernel_version must be 0 (auto), 1, 2, 3, 4 or 5");
        ctx->opt_kernel = v;
    }
    else
        return fail(ctx, PCL_EINVAL, "unknown option '%s'", key);
    return PCL_OK;
}
#ifdef PCL_LAB  // include/piccolo_hip_lab.h; the stamps themselves need -DPCL_PROFILE as well
extern "C" int pcl_debug_timing(pcl_ctx *ctx, int64_t *out, int64_t cap) {
    if (!ctx || !out || cap < 0) return PCL_EINVAL;
    if (!ctx->ddbg) return fail(ctx, PCL_EINVAL, "pcl_debug_timing: set option debug_timing first");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out, ctx->ddbg, (size_t)std::min<int64_t>(cap, PCL_DBG_WORDS) * sizeof(long long), hipMemcpyDeviceToHost));
    return PCL_OK;
}
#endif

extern "C" int pcl_get_option(const pcl_ctx *ctx, const char *key, int64_t *v) {
    if (!ctx || !key || !v) return PCL_EINVAL;
    if (!strcmp(key, "cols_per_slice"))
        *v = ctx->opt_cols_per_slice;
    else if (!strcmp(key, "use_mfma"))
        *v = ctx->opt_use_mfma;
    else if (!strcmp(key, "nt_stores"))
        *v = ctx->opt_nt;
    else if (!strcmp(key, "effective_cols_per_slice"))
        *v = ((ctx->opt_kernel == 3 || (ctx->opt_kernel == 0 && v3_specialised(ctx))) && ctx->opt_use_mfma && v3_supported(ctx) && !ctx->vec && ctx->cols == ctx->desc.d)
                 ? (v3_contiguous(ctx) ? ctx->desc.d : choose_cols_v3(ctx))
                 : choose_cols_per_slice(ctx, true);
    else if (!strcmp(key, "n_cu"))
        *v = ctx->n_cu;
    else if (!strcmp(key, "host_threads"))
        *v = host_threads(ctx);
    else if (!strcmp(key, "host_expand_MBps"))  // delivered rate of the fastest call of the thread-count sweep (0 before it has finished)
        *v = (int64_t)(ctx->host_expand_GBps * 1e3);
    else if (!strcmp(key, "host_path"))
        *v = ctx->opt_host_path;
    else if (!strcmp(key, "host_chunks"))
        *v = ctx->opt_host_chunks;
    else if (!strcmp(key, "kernel_version"))
        *v = ctx->opt_kernel;
    else if (!strcmp(key, "last_kernel"))
        *v = ctx->last_kernel;
    else if (!strcmp(key, "hess_xcd"))
        *v = ctx->opt_hess_xcd;
    else if (!strcmp(key, "hess_rpre"))
        *v = ctx->opt_hess_rpre;
    else if (!strcmp(key, "last_hess_rpre"))
        *v = ctx->last_hess_rpre;
    else if (!strcmp(key, "hess_pair"))
        *v = ctx->opt_hess_pair;
    else if (!strcmp(key, "last_hess_pair"))
        *v = ctx->last_hess_pair;
#ifdef PCL_LAB
    else if (!strcmp(key, "resident_idle_us"))
        *v = ctx->opt_resident_idle_us;
    else if (!strcmp(key, "resident_launches"))  // starts of the resident kernel since pcl_create (1 + the times it had left when a request came)
        *v = ctx->res.launches;
#endif
    else if (!strcmp(key, "v4_tune"))
        *v = ctx->opt_v4_tune;
    else if (!strcmp(key, "last_v4_tune_choice"))  // of the array the last multi-trajectory launch wrote: -1 still sampling | 0 static split | 1 slice tickets
        *v = ctx->last_v4_tune_choice;
    else if (!strcmp(key, "last_v4_tune_static_ns"))  // best timed launch of each variant on the last decided array
        *v = ctx->last_v4_tune_static_ns;
    else if (!strcmp(key, "last_v4_tune_ticket_ns"))
        *v = ctx->last_v4_tune_ticket_ns;
    else if (!strcmp(key, "host_store_bytes"))  // the width the last host expansion used
        *v = ctx->last_host_store_bytes;
    else if (!strcmp(key, "cgroup_quota_cpus_x100"))  // 100 x the CPUs the cgroup grants the process (cpu.max); 0: no quota
        *v = (int64_t)(pcl_host::cgroup_quota_cpus() * 100.0 + 0.5);
    else if (!strcmp(key, "last_hess_split"))
        *v = ctx->last_hess_split;
    else if (!strcmp(key, "last_eval_coop"))
        *v = ctx->last_eval_coop;
    else if (!strcmp(key, "last_objective_launches"))
        *v = ctx->last_objective_launches;
    else if (!strcmp(key, "objective_launches"))
        *v = ctx->opt_objective_launches;
    else if (!strcmp(key, "last_step_launches"))  // pcl_eval_jac_merit_objective_dev: 2 = fused kernel + one tail launch, 4 = the separate calls
        *v = ctx->last_step_launches;
    else if (!strcmp(key, "last_merit_fused"))
        *v = ctx->merit_fused;
    else if (!strcmp(key, "contiguous"))
        *v = ctx->opt_contig;
    else if (!strcmp(key, "jit"))
        *v = ctx->opt_jit;
    else if (!strcmp(key, "jit_compiles")) {  // modules this process compiled with hiprtc
        std::lock_guard<std::mutex> lock(g_jit_mutex);
        *v = g_jit_compiles;
    } else if (!strcmp(key, "jit_cache_hits")) {  // ... and loaded from the prebuilt directory or the on-disk cache instead
        std::lock_guard<std::mutex> lock(g_jit_mutex);
        *v = g_jit_cache_hits;
    } else if (!strcmp(key, "jit_fallbacks"))  // pattern-compiled kernels this context wanted and did not get
        *v = ctx->jit_fallbacks;
    else if (!strcmp(key, "require_jit"))
        *v = ctx->opt_require_jit;
    else if (!strcmp(key, "pade_order"))  // the order in use (0: pade_order = 0 at creation and nothing has chosen yet)
        *v = ctx->desc.pade_order;
    else if (!strcmp(key, "order_tol_met"))  // 1 unless the order policy settled for order 10 with its bound above the tolerance
        *v = ctx->order_tol_met;
    else if (!strcmp(key, "order_theta_1e9"))  // 1e9 x the bound on |dt G|_2 the order policy worked with
        *v = (int64_t)(ctx->order_theta * 1e9);
    else if (!strcmp(key, "stream_workgroups"))
        *v = ctx->opt_stream_wg;
    else if (!strcmp(key, "last_stream_workgroups"))
        *v = ctx->last_n_stream;
    else if (!strcmp(key, "v4_ticket"))
        *v = ctx->opt_v4_ticket;
    else if (!strcmp(key, "v4_ticket_cols"))
        *v = ctx->opt_v4_ticket_cols;
    else if (!strcmp(key, "v4_group"))
        *v = ctx->opt_v4_group;
    else if (!strcmp(key, "v4_ticket_ahead"))
        *v = ctx->opt_v4_ticket_ahead;
    else if (!strcmp(key, "last_v4_ticket"))
        *v = ctx->last_v4_ticket;
    else if (!strcmp(key, "hess_kernel"))
        *v = ctx->opt_hess_kernel;
    else if (!strcmp(key, "eval_kernel"))
        *v = ctx->opt_eval_kernel;
    else if (!strcmp(key, "last_hess_kernel"))
        *v = ctx->last_hess_kernel;
    else if (!strcmp(key, "ell_width_t"))
        *v = ctx->ellt_w;
    else if (!strcmp(key, "drives_antisymmetric"))
        *v = ctx->drives_antisym;
    else if (!strcmp(key, "occupancy_v2")) {
        KParams p;
        fill_params(ctx, p);
        p.nc = choose_cols_per_slice(ctx, true);
        const size_t lds = fused2_lds_bytes(p, true, ell_fits_lds(ctx));
        int nb = 0;
        (void)hipFuncSetAttribute((const void *)pcl_fused_kernel_v2<true, 1, 0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pcl_fused_kernel_v2<true, 1, 0, 0, 0>, 512, lds) != hipSuccess) nb = -1;
        *v = nb * 1000000LL + (long long)lds;
    }
    else if (!strcmp(key, "iso_structured"))
        *v = ctx->iso;
    else if (!strcmp(key, "ell_width"))
        *v = ctx->ell_w;
    else if (!strcmp(key, "union_width"))
        *v = ctx->uell_w;
    else
        return fail(ctx, PCL_EINVAL, "unknown option '%s'", key);
    return PCL_OK;
}

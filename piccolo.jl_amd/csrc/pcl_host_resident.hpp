// pcl_host_resident.hpp -- part of piccolo_hip.hip, included there ONLY in lab builds (-DPCL_LAB; include/piccolo_hip_lab.h): the RESIDENT
// evaluator of round 5.  Kernel 4's workgroups stay on the device and run one evaluation per posted request.  It was built because a review asked
// for the number, it is bitwise equal to the launched kernel, and it LOSES (32-33 us per evaluation against 24.2 for launches: DESIGN.md 4.2.2,
// lab/probes/resident_probe.py), so it is not part of the shipped library or of its ABI.
#pragma once

// ---- resident evaluator ------------------------------------------------------------------------------------------------------------
static bool res_running(pcl_ctx *ctx) {
    if (!ctx->res.launched) return false;
    const hipError_t e = hipStreamQuery(ctx->res.stream);
    (void)hipGetLastError();
    return e == hipErrorNotReady;
}
// (re)start the kernel at the first evaluation that is not complete; requests already posted stay posted
static int res_launch(pcl_ctx *ctx) {
    pcl_ctx::Resident &R = ctx->res;
    HIP_TRY(ctx, hipStreamSynchronize(R.stream));  // (the previous resident kernel has left: its last words are written)
    const unsigned done = __atomic_load_n(R.hbox + 16, __ATOMIC_ACQUIRE);
    __atomic_store_n(R.hbox + 1, 0u, __ATOMIC_RELEASE);
    __atomic_store_n(R.hbox + 17, 0u, __ATOMIC_RELEASE);
    memset(R.hinit, 0, 64 * sizeof(unsigned));
    R.hinit[0] = done, R.hinit[2] = done;
    R.hinit[40] = (unsigned)((unsigned long long)R.hbox_dev & 0xffffffffu), R.hinit[41] = (unsigned)((unsigned long long)R.hbox_dev >> 32);
    R.hinit[42] = done;
    R.hinit[43] = (unsigned)std::min<int64_t>(std::max<int64_t>(ctx->opt_resident_idle_us, 10), 2000000) * 100u;  // 100 MHz ticks
    R.hinit[44] = done + (1u << 30);
    HIP_TRY(ctx, hipMemcpyAsync(R.dbox, R.hinit, 64 * sizeof(unsigned), hipMemcpyHostToDevice, R.stream));
    *R.hparams = R.p;
    HIP_TRY(ctx, hipMemcpyAsync(R.dparams, R.hparams, sizeof(KParams), hipMemcpyHostToDevice, R.stream));
    void *args[] = {(void *)&R.dparams, (void *)&R.tab, (void *)&ctx->dv4_mags, (void *)&R.dcf, (void *)&R.dbox};
    HIP_TRY(ctx, hipModuleLaunchKernel(R.f, (unsigned)R.grid, 1, 1, R.block, 1, 1, (unsigned)(R.lds + 32), R.stream, args, nullptr));
    R.launched = true;
    ++R.launches;
    return PCL_OK;
}
extern "C" int pcl_resident_start(pcl_ctx *ctx, const double *Z, double *delta, double *vals) {
    if (!ctx) return PCL_EINVAL;
    if (!Z || !vals) return fail(ctx, PCL_EINVAL, "pcl_resident_start: NULL pointer");
    if (ctx->res.active) return fail(ctx, PCL_EINVAL, "pcl_resident_start: already started (pcl_resident_stop first)");
    ON_DEVICE(ctx);
    pcl_ctx::Resident &R = ctx->res;
    ctx->res_capture = true;
    const int rc = launch_fused(ctx, Z, delta, vals, false);
    ctx->res_capture = false;
    if (rc != PCL_OK) return rc;
    if (R.lds + 32 > (size_t)ctx->max_lds) return fail(ctx, PCL_ESHAPE, "pcl_resident_start: no LDS word left for the request flag");
    if (R.grid > std::max(ctx->n_cu, 1)) return fail(ctx, PCL_ESHAPE, "pcl_resident_start: more workgroups than CUs");
    if (!R.f) {
        const int np = v4_power_tiles(R.p.d, R.p.m, R.p.q, (size_t)ctx->max_lds);
        const std::string src = v4_source(*ctx->v4_plan, R.p.q, np, (int)ctx->opt_v4_variant, 2);
        const std::string key = "fused-sparse-resident:" + std::to_string(R.p.q) + ":" + std::to_string(std::hash<std::string>{}(src));
        R.f = jit_compile(ctx->device, key, src, "pcl_fused_sparse_resident", true);
        if (!R.f) return fail(ctx, PCL_EHIP, "pcl_resident_start: the resident module did not compile (%s)", g_jit_note.c_str());
    }
    if (!R.stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
    if (!R.hbox) {
        HIP_TRY(ctx, hipHostMalloc((void **)&R.hbox, 64 * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(ctx, hipHostGetDevicePointer((void **)&R.hbox_dev, R.hbox, 0));
        HIP_TRY(ctx, hipHostMalloc((void **)&R.hinit, 64 * sizeof(unsigned), hipHostMallocDefault));
        HIP_TRY(ctx, hipHostMalloc((void **)&R.hparams, sizeof(KParams), hipHostMallocDefault));
        HIP_TRY(ctx, hipMalloc((void **)&R.dparams, sizeof(KParams)));
        HIP_TRY(ctx, hipMalloc((void **)&R.dbox, 64 * sizeof(unsigned) + 16 * 256 * 4 * sizeof(long long)));  // (+ the debugging stamps of v4_flags & 2048)
        HIP_TRY(ctx, hipMemset(R.dbox, 0, 64 * sizeof(unsigned) + 16 * 256 * 4 * sizeof(long long)));
    }
    for (int i = 0; i < 64; ++i) __atomic_store_n(R.hbox + i, 0u, __ATOMIC_RELAXED);
    R.posted = 0;
    R.launched = false;
    // what the trajectory and the outputs' earlier writers have queued on the context's stream comes first
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int rl = res_launch(ctx);
    R.active = rl == PCL_OK;  // (a start that failed leaves the context as it was: it can be started again)
    return rl;
}
extern "C" int pcl_resident_post(pcl_ctx *ctx, int32_t count) {
    if (!ctx) return PCL_EINVAL;
    if (!ctx->res.active) return fail(ctx, PCL_EINVAL, "pcl_resident_post: not started");
    if (count < 1 || count > (1 << 20)) return fail(ctx, PCL_EINVAL, "pcl_resident_post: count %d", (int)count);
    ON_DEVICE(ctx);
    pcl_ctx::Resident &R = ctx->res;
    R.posted += (unsigned)count;
    __atomic_store_n(R.hbox + 0, R.posted, __ATOMIC_RELEASE);
    // (it has left -- idle for longer than resident_idle_us -- or is leaving: workgroup 0 says so in a host word; no runtime call on the way of a request)
    if (__atomic_load_n(R.hbox + 17, __ATOMIC_ACQUIRE)) return res_launch(ctx);
    return PCL_OK;
}
extern "C" int pcl_resident_wait(pcl_ctx *ctx, double timeout_s) {
    if (!ctx) return PCL_EINVAL;
    if (!ctx->res.active) return fail(ctx, PCL_EINVAL, "pcl_resident_wait: not started");
    ON_DEVICE(ctx);
    pcl_ctx::Resident &R = ctx->res;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long long spin = 0;; ++spin) {
        if (__atomic_load_n(R.hbox + 16, __ATOMIC_ACQUIRE) == R.posted) break;
        if ((spin & 1023) == 1023) {
            if (!res_running(ctx)) {  // left with requests outstanding (it decided to leave as they arrived): again from the first incomplete one
                if (__atomic_load_n(R.hbox + 16, __ATOMIC_ACQUIRE) == R.posted) break;
                if (int rc = res_launch(ctx)) return rc;
            }
            if (timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                __atomic_store_n(R.hbox + 1, 1u, __ATOMIC_RELEASE);
                return fail(ctx, PCL_EINTERNAL, "pcl_resident_wait: %u of %u requests complete after %.3f s (the kernel has been told to leave)", __atomic_load_n(R.hbox + 16, __ATOMIC_ACQUIRE), R.posted, timeout_s);
            }
        }
    }
    return check_device_error(ctx, "pcl_resident_wait");
}
extern "C" int pcl_resident_stop(pcl_ctx *ctx) {
    if (!ctx) return PCL_EINVAL;
    pcl_ctx::Resident &R = ctx->res;
    if (!R.active) return PCL_OK;
    ON_DEVICE(ctx);
    __atomic_store_n(R.hbox + 1, 1u, __ATOMIC_RELEASE);
    R.active = false;
    HIP_TRY(ctx, hipStreamSynchronize(R.stream));
    R.launched = false;
    return check_device_error(ctx, "pcl_resident_stop");
}
// debugging (option v4_flags & 2048 at pcl_resident_start): 100 MHz stamps [evaluation 0 .. 15 since the last start][workgroup][request seen, last block
// store issued, workgroup drained, arrival counted]; call after pcl_resident_stop
extern "C" int pcl_resident_stamps(pcl_ctx *ctx, int64_t *out, int64_t count) {
    if (!ctx || !out || !ctx->res.dbox) return PCL_EINVAL;
    ON_DEVICE(ctx);
    HIP_TRY(ctx, hipMemcpy(out, ctx->res.dbox + 64, (size_t)std::min<int64_t>(count, 16 * 256 * 4) * sizeof(int64_t), hipMemcpyDeviceToHost));
    return PCL_OK;
}
extern "C" int pcl_resident_completed(const pcl_ctx *ctx, int64_t *count) {
    if (!ctx || !count) return PCL_EINVAL;
    *count = ctx->res.hbox ? (int64_t)__atomic_load_n(ctx->res.hbox + 16, __ATOMIC_ACQUIRE) : -1;
    return PCL_OK;
}

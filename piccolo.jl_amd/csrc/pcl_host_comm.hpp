// pcl_host_comm.hpp -- part of piccolo_hip.hip (included there, in place): the RCCL sum-reduce behind the C ABI (librccl opened lazily).
#pragma once
// --- RCCL sum-reduce of the shared-control payload -----------------------------
// librccl is opened lazily with dlopen so that single-GPU users never load it.  ncclUniqueId is 128 opaque bytes;
// ncclDataType_t ncclFloat64 = 8, ncclRedOp_t ncclSum = 0 (rccl.h of ROCm 7.x).
namespace {
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, pcl_comm_id, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load(const pcl_ctx *ctx) {
    if (g_rccl.h) return PCL_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(ctx, PCL_ERCCL, "dlopen(librccl.so): %s", dlerror());
    RcclApi a;
    a.h = h;
    a.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, pcl_comm_id, int))dlsym(h, "ncclCommInitRank");
    a.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
    a.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    a.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy) return fail(ctx, PCL_ERCCL, "librccl lacks the expected symbols");
    g_rccl = a;
    return PCL_OK;
}
int rccl_fail(const pcl_ctx *ctx, const char *what, int rc) {
    return fail(ctx, PCL_ERCCL, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
}
}  // namespace

extern "C" int pcl_comm_get_unique_id(pcl_comm_id *out) {
    if (!out) return fail(nullptr, PCL_EINVAL, "pcl_comm_get_unique_id: NULL");
    int rc = rccl_load(nullptr);
    if (rc != PCL_OK) return rc;
    int nrc = g_rccl.GetUniqueId(out);
    return nrc == 0 ? PCL_OK : rccl_fail(nullptr, "ncclGetUniqueId", nrc);
}
extern "C" int pcl_comm_init(pcl_ctx *ctx, const pcl_comm_id *id, int32_t rank, int32_t nranks) {
    if (!ctx || !id) return PCL_EINVAL;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, PCL_EINVAL, "pcl_comm_init: rank %d of %d", rank, nranks);
    if (ctx->comm) return fail(ctx, PCL_EINVAL, "pcl_comm_init: communicator already initialised");
    TRY(rccl_load(ctx));
    ON_DEVICE(ctx);
    int nrc = g_rccl.CommInitRank(&ctx->comm, nranks, *id, rank);
    if (nrc != 0) {
        ctx->comm = nullptr;
        return rccl_fail(ctx, "ncclCommInitRank", nrc);
    }
    return PCL_OK;
}
extern "C" int pcl_reduce_sum_dev(pcl_ctx *ctx, double *buf_dev, int64_t n) {
    if (!ctx) return PCL_EINVAL;
    if (!buf_dev || n < 0) return fail(ctx, PCL_EINVAL, "pcl_reduce_sum_dev: bad buffer");
    if (!ctx->comm) return fail(ctx, PCL_ERCCL, "pcl_reduce_sum_dev: call pcl_comm_init first");
    ON_DEVICE(ctx);
    int nrc = g_rccl.AllReduce(buf_dev, buf_dev, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    return nrc == 0 ? PCL_OK : rccl_fail(ctx, "ncclAllReduce", nrc);
}
extern "C" int pcl_reduce_sum(pcl_ctx *ctx, double *buf, int64_t n) {  // host buffer, staged through device memory; synchronous
    if (!ctx) return PCL_EINVAL;
    if (!buf || n < 0) return fail(ctx, PCL_EINVAL, "pcl_reduce_sum: bad buffer");
    if (!ctx->comm) return fail(ctx, PCL_ERCCL, "pcl_reduce_sum: call pcl_comm_init first");
    if (n == 0) return PCL_OK;
    ON_DEVICE(ctx);
    if (ctx->reduce_cap < n) {
        if (ctx->dreduce) (void)hipFree(ctx->dreduce);
        ctx->dreduce = nullptr;
        ctx->reduce_cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->dreduce, (size_t)n * sizeof(double)));
        ctx->reduce_cap = n;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dreduce, buf, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = pcl_reduce_sum_dev(ctx, ctx->dreduce, n)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(buf, ctx->dreduce, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PCL_OK;
}
extern "C" int pcl_comm_destroy(pcl_ctx *ctx) {
    if (!ctx) return PCL_EINVAL;
    if (ctx->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    return PCL_OK;
}

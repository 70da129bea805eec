// bench_abi.cpp -- a C++ client of the drop-in boundary: include/piccolo_hip.h and the HIP runtime, nothing else (no Python, no
// torch).  Times BASELINE's metric -- fused residual + Jacobian evaluations per second of ONE config-3 trajectory, inputs and
// outputs resident in HBM -- through exactly the calls a host binding makes (pcl_create, pcl_jac_structure, pcl_eval_jac_dev,
// pcl_eval_jac) and checks the device-pointer path against the host-pointer path bit for bit.
//   python bench/make_abi_inputs.py                       # writes bench/config3_inputs.bin (numpy only)
//   hipcc -O2 -std=c++17 -I include -o bench/bench_abi bench/bench_abi.cpp -L piccolo.jl_amd/csrc -lpiccolo_hip -Wl,-rpath,'$ORIGIN/../piccolo.jl_amd/csrc'
//   bench/bench_abi [inputs.bin] [steps] [warmup]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "piccolo_hip.h"

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                 \
            return 2;                                                                               \
        }                                                                                           \
    } while (0)
#define PCLCHK(x)                                                                                   \
    do {                                                                                            \
        int r_ = (x);                                                                               \
        if (r_ != PCL_OK) {                                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #x, r_, pcl_last_error(ctx));                         \
            return 3;                                                                               \
        }                                                                                           \
    } while (0)

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "bench/config3_inputs.bin";
    const int steps = argc > 2 ? atoi(argv[2]) : 200, warmup = argc > 3 ? atoi(argv[3]) : 20;
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s (run: python bench/make_abi_inputs.py)\n", path);
        return 1;
    }
    int32_t hdr[8];
    if (fread(hdr, sizeof hdr, 1, f) != 1) return 1;
    const int d = hdr[0], m = hdr[1], N = hdr[2], z_dim = hdr[3], n = 2 * d;
    const int32_t x_off = hdr[4];
    std::vector<double> G0((size_t)n * n), Gj((size_t)m * n * n), Z((size_t)N * z_dim);
    if (fread(G0.data(), 8, G0.size(), f) != G0.size() || fread(Gj.data(), 8, Gj.size(), f) != Gj.size() || fread(Z.data(), 8, Z.size(), f) != Z.size()) return 1;
    fclose(f);

    pcl_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.struct_size = (int32_t)sizeof desc;
    desc.d = d;
    desc.n_drives = m;
    desc.N = N;
    desc.z_dim = z_dim;
    desc.u_off = hdr[5];
    desc.dt_off = hdr[6];
    desc.batch = 1;
    desc.batch_mode = PCL_BATCH_MEMBERS;
    desc.pade_order = 4;
    desc.index_base = 1;  // what a Julia / MOI host asks for
    desc.G0 = G0.data();
    desc.Gj = Gj.data();
    desc.x_offs = &x_off;
    pcl_ctx *ctx = nullptr;
    if (pcl_create(&desc, &ctx) != PCL_OK) {
        fprintf(stderr, "pcl_create: %s\n", pcl_last_error(nullptr));
        return 3;
    }
    int64_t x_dim = 0, n_rows = 0, n_cols = 0, nnz = 0, per = 0;
    PCLCHK(pcl_constraint_dim(ctx, &x_dim, &n_rows, &n_cols));
    PCLCHK(pcl_jac_nnz(ctx, &nnz, &per));
    std::vector<int32_t> rows((size_t)nnz), cols((size_t)nnz);
    PCLCHK(pcl_jac_structure(ctx, rows.data(), cols.data()));
    int32_t rmax = 0, cmax = 0;
    for (int64_t i = 0; i < nnz; ++i) {
        rmax = rows[i] > rmax ? rows[i] : rmax;
        cmax = cols[i] > cmax ? cols[i] : cmax;
    }
    printf("%s | d %d m %d N %d | rows %lld cols %lld nnz %lld (%lld per interval) | structure 1-based, max (row, col) = (%d, %d)\n", pcl_version(), d, m, N,
           (long long)n_rows, (long long)n_cols, (long long)nnz, (long long)per, rmax, cmax);
    if (rmax != n_rows || cmax > n_cols) {
        fprintf(stderr, "structure out of range\n");
        return 4;
    }

    hipStream_t stream;
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    PCLCHK(pcl_set_stream(ctx, (void *)stream));
    double *dZ, *dd, *dv;
    HIPCHK(hipMalloc((void **)&dZ, Z.size() * 8));
    HIPCHK(hipMalloc((void **)&dd, (size_t)n_rows * 8));
    HIPCHK(hipMalloc((void **)&dv, (size_t)nnz * 8));
    HIPCHK(hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
    for (int i = 0; i < warmup; ++i) PCLCHK(pcl_eval_jac_dev(ctx, dZ, dd, dv));
    HIPCHK(hipStreamSynchronize(stream));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventRecord(e0, stream));
    for (int i = 0; i < steps; ++i) PCLCHK(pcl_eval_jac_dev(ctx, dZ, dd, dv));
    HIPCHK(hipEventRecord(e1, stream));
    HIPCHK(hipStreamSynchronize(stream));
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = ((double)z_dim * 8 + (double)x_dim * 8 + (double)per * 8) * (N - 1);  // algorithmic bytes per evaluation

    // the host-pointer entry point (what the Julia glue calls) must deliver the same bits
    std::vector<double> hd((size_t)n_rows), hv((size_t)nnz), gd((size_t)n_rows), gv((size_t)nnz);
    PCLCHK(pcl_reset_stream(ctx));
    PCLCHK(pcl_eval_jac(ctx, Z.data(), hd.data(), hv.data()));
    HIPCHK(hipMemcpy(gd.data(), dd, gd.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(gv.data(), dv, gv.size() * 8, hipMemcpyDeviceToHost));
    const bool same = !memcmp(hd.data(), gd.data(), gd.size() * 8) && !memcmp(hv.data(), gv.data(), gv.size() * 8);
    double dmax = 0.0;
    for (double v : gd) dmax = fabs(v) > dmax ? fabs(v) : dmax;
    const auto h0 = std::chrono::steady_clock::now();
    const int hreps = 10;
    for (int i = 0; i < hreps; ++i) PCLCHK(pcl_eval_jac(ctx, Z.data(), hd.data(), hv.data()));
    const double hwall = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
    int64_t lk = 0;
    PCLCHK(pcl_get_option(ctx, "last_kernel", &lk));
    printf("{\"client\": \"bench_abi.cpp (C ABI only)\", \"metric\": \"constraint+Jacobian evals/sec, 3-transmon d=27 unitary, T=100 knots\", \"value\": %.1f, "
           "\"unit\": \"evals/s\", \"steps\": %d, \"warmup\": %d, \"us_per_eval_wall\": %.2f, \"us_per_eval_kernel\": %.2f, \"hbm_GBps\": %.1f, "
           "\"host_delivered_evals_per_s\": %.1f, \"device_and_host_paths_bitwise_equal\": %s, \"max_abs_delta\": %.3e, \"last_kernel\": %lld}\n",
           steps / wall, steps, warmup, wall / steps * 1e6, ms * 1e3 / steps, bytes / (ms * 1e-3 / steps) / 1e9, hreps / hwall, same ? "true" : "false", dmax, (long long)lk);
    pcl_destroy(ctx);
    (void)hipFree(dZ);
    (void)hipFree(dd);
    (void)hipFree(dv);
    return same && std::isfinite(dmax) && dmax > 0.0 ? 0 : 5;
}

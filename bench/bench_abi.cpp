// bench_abi.cpp -- a C++ client of the drop-in boundary: include/piccolo_hip.h and the HIP runtime, nothing else (no Python, no
// torch).  Times BASELINE's metric -- fused residual + Jacobian evaluations per second of ONE config-3 trajectory, inputs and
// outputs resident in HBM -- through exactly the calls a host binding makes (pcl_create, pcl_jac_structure, pcl_eval_jac_dev,
// pcl_eval_jac) and checks the device-pointer path against the host-pointer path bit for bit.
//   python bench/make_abi_inputs.py                       # writes bench/config3_inputs.bin (numpy only)
//   hipcc -O2 -std=c++17 -I include -o bench/bench_abi bench/bench_abi.cpp -L piccolo.jl_amd/csrc -lpiccolo_hip -Wl,-rpath,'$ORIGIN/../piccolo.jl_amd/csrc'
//   bench/bench_abi [inputs.bin] [steps] [warmup] [--ranks N]
// --ranks N (N >= 1): the path's ONE collective through the C ABI as well -- pcl_comm_get_unique_id (rank 0; the 128 bytes travel through a file) ->
// pcl_comm_init -> pcl_reduce_sum_dev of a payload of the shared controls' size (1 + (N-1)(m+1) doubles) behind every evaluation ->
// pcl_comm_destroy.  One PROCESS per device: the parent is rank 0 and starts ranks 1 .. N-1 as copies of itself (--rank r --id-file f), rank r on
// device r.  N = 1 runs the same calls over one rank (what a 1-GPU box can execute); the sum is checked on every rank: rank r contributes r + 1.
#include <hip/hip_runtime.h>

#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
    int nranks = 0, rank = 0;  // nranks 0: no collective (the default)
    const char *id_file = nullptr;
    std::vector<char *> pos;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--ranks") && i + 1 < argc)
            nranks = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--rank") && i + 1 < argc)
            rank = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--id-file") && i + 1 < argc)
            id_file = argv[++i];
        else
            pos.push_back(argv[i]);
    }
    const char *path = pos.size() > 0 ? pos[0] : "bench/config3_inputs.bin";
    const int steps = pos.size() > 1 ? atoi(pos[1]) : 200, warmup = pos.size() > 2 ? atoi(pos[2]) : 20;
    if (nranks < 0 || rank < 0 || (nranks > 0 && rank >= nranks)) {
        fprintf(stderr, "--ranks %d --rank %d\n", nranks, rank);
        return 1;
    }
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (nranks > n_dev) {
        fprintf(stderr, "--ranks %d asked for, %d device(s) visible\n", nranks, n_dev);
        return 1;
    }
    // rank 0 of a multi-rank run: the id (written to a file the children poll), then the children
    pcl_comm_id comm_id;
    std::string idf;
    std::vector<pid_t> kids;
    if (nranks > 0 && rank == 0) {
        if (pcl_comm_get_unique_id(&comm_id) != PCL_OK) {
            fprintf(stderr, "pcl_comm_get_unique_id: %s\n", pcl_last_error(nullptr));
            return 3;
        }
        if (nranks > 1) {
            idf = std::string("/tmp/bench_abi_id_") + std::to_string((long long)getpid());
            FILE *g = fopen((idf + ".tmp").c_str(), "wb");
            if (!g || fwrite(&comm_id, sizeof comm_id, 1, g) != 1) return 1;
            fclose(g);
            if (rename((idf + ".tmp").c_str(), idf.c_str()) != 0) return 1;
            for (int r = 1; r < nranks; ++r) {
                const pid_t pid = fork();
                if (pid == 0) {
                    const std::string rs = std::to_string(r), ns = std::to_string(nranks), st = std::to_string(steps), wu = std::to_string(warmup);
                    execl("/proc/self/exe", argv[0], path, st.c_str(), wu.c_str(), "--ranks", ns.c_str(), "--rank", rs.c_str(), "--id-file", idf.c_str(), (char *)nullptr);
                    _exit(127);
                }
                kids.push_back(pid);
            }
        }
    } else if (nranks > 0) {
        if (!id_file) return 1;
        FILE *g = nullptr;
        for (int tries = 0; tries < 600 && !(g = fopen(id_file, "rb")); ++tries) usleep(100000);
        if (!g || fread(&comm_id, sizeof comm_id, 1, g) != 1) return 1;
        fclose(g);
    }
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
    desc.device_id = nranks > 0 ? rank : 0;  // one process per device
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
    if (rank == 0) printf("%s | d %d m %d N %d | rows %lld cols %lld nnz %lld (%lld per interval) | structure 1-based, max (row, col) = (%d, %d)\n", pcl_version(), d, m, N,
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

    // the path's one collective through the C ABI: a payload of the shared controls' size summed over the ranks behind every evaluation
    double red_us = -1.0, alone_us = -1.0;
    bool red_ok = true;
    const int64_t npay = 1 + (int64_t)(N - 1) * (m + 1);
    if (nranks > 0) {
        PCLCHK(pcl_comm_init(ctx, &comm_id, rank, nranks));
        double *dp;
        HIPCHK(hipMalloc((void **)&dp, (size_t)npay * 8));
        std::vector<double> hp((size_t)npay, (double)(rank + 1));
        auto reset_payload = [&]() { return hipMemcpyAsync(dp, hp.data(), (size_t)npay * 8, hipMemcpyHostToDevice, stream); };
        HIPCHK(reset_payload());
        PCLCHK(pcl_reduce_sum_dev(ctx, dp, npay));  // (the first one builds the rings)
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<double> got((size_t)npay);
        HIPCHK(hipMemcpy(got.data(), dp, (size_t)npay * 8, hipMemcpyDeviceToHost));
        const double want = 0.5 * nranks * (nranks + 1);  // sum over ranks of (rank + 1)
        for (double v : got) red_ok = red_ok && v == want;
        HIPCHK(hipEventRecord(e0, stream));
        for (int i = 0; i < steps; ++i) {
            PCLCHK(pcl_eval_jac_dev(ctx, dZ, dd, dv));
            PCLCHK(pcl_reduce_sum_dev(ctx, dp, npay));
        }
        HIPCHK(hipEventRecord(e1, stream));
        HIPCHK(hipStreamSynchronize(stream));
        float rms = 0.f;
        HIPCHK(hipEventElapsedTime(&rms, e0, e1));
        red_us = rms * 1e3 / steps;
        HIPCHK(hipEventRecord(e0, stream));
        for (int i = 0; i < steps; ++i) PCLCHK(pcl_reduce_sum_dev(ctx, dp, npay));
        HIPCHK(hipEventRecord(e1, stream));
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipEventElapsedTime(&rms, e0, e1));
        alone_us = rms * 1e3 / steps;
        PCLCHK(pcl_comm_destroy(ctx));
        (void)hipFree(dp);
    }

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
    if (rank == 0) printf("{\"client\": \"bench_abi.cpp (C ABI only)\", \"metric\": \"constraint+Jacobian evals/sec, 3-transmon d=27 unitary, T=100 knots\", \"value\": %.1f, "
           "\"unit\": \"evals/s\", \"steps\": %d, \"warmup\": %d, \"us_per_eval_wall\": %.2f, \"us_per_eval_kernel\": %.2f, \"hbm_GBps\": %.1f, "
           "\"host_delivered_evals_per_s\": %.1f, \"device_and_host_paths_bitwise_equal\": %s, \"max_abs_delta\": %.3e, \"last_kernel\": %lld}\n",
           steps / wall, steps, warmup, wall / steps * 1e6, ms * 1e3 / steps, bytes / (ms * 1e-3 / steps) / 1e9, hreps / hwall, same ? "true" : "false", dmax, (long long)lk);
    pcl_destroy(ctx);
    (void)hipFree(dZ);
    (void)hipFree(dd);
    (void)hipFree(dv);
    int rc = same && std::isfinite(dmax) && dmax > 0.0 ? 0 : 5;
    if (nranks > 0) {
        if (!red_ok) rc = 6;
        int kids_ok = 1;
        for (pid_t pid : kids) {
            int st = 0;
            if (waitpid(pid, &st, 0) != pid || !WIFEXITED(st) || WEXITSTATUS(st) != 0) kids_ok = 0;
        }
        if (!idf.empty()) (void)unlink(idf.c_str());
        if (!kids_ok) rc = rc ? rc : 7;
        if (rank == 0)  // (one more line, after the evaluation's: the collective)
            printf("{\"client\": \"bench_abi.cpp (C ABI only)\", \"rccl_ranks\": %d, \"payload_doubles\": %lld, \"sum_exact_on_rank0\": %s, \"other_ranks_ok\": %s, "
                   "\"us_per_eval_plus_reduce\": %.2f, \"us_per_reduce_alone\": %.2f, \"evals_per_s_all_ranks\": %.1f}\n",
                   nranks, (long long)npay, red_ok ? "true" : "false", kids_ok ? "true" : "false", red_us, alone_us, nranks * 1e6 / red_us);
    }
    return rc;
}

// Round 5: why does the replicating expander (compact tiles -> full values) run at 2.9 TB/s when a fill of the same bytes runs at 6.7?
// Variants of ONE short-lived-workgroup kernel on the exact record geometry of config 3 (per interval: 27 copies of a 54x54 tile, 27 copies
// of another, a 10,206-double tail): source = registers (no loads) | one small L2-resident tile | the real compact array; slices of cpi columns;
// block order; 256 or 512 threads.  hipMemsetAsync on the same buffer beside it.
//   hipcc --offload-arch=gfx950 -O3 -o lab/probes/expand_probe lab/probes/expand_probe.hip ; expand_probe [trajectories=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int D = 27, N = 54, NN = N * N, M = 6;
constexpr long long XD = (long long)N * D, CPER = 2 * NN + XD * (M + 1), FPER = 2LL * D * NN + XD * (M + 1);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// src_mode 0: registers | 1: every workgroup reads the SAME tile (L2-resident) | 2: its interval's tile of the compact array
// order 0: per interval [sign 0 slices | sign 1 slices | tails] | 1: all intervals' block work first, tails behind
template <int NT>
__global__ __launch_bounds__(NT) void expand(const double *__restrict__ compact, double *__restrict__ full, long long n_bk, int cpi, int src_mode, int sc) {
    const int S = (D + cpi - 1) / cpi;
    constexpr int TP = 2048;  // pairs per tail workgroup
    const long long tail2 = (XD * (M + 1)) >> 1;
    const int T = (int)((tail2 + TP - 1) / TP);
    const int per_bk = 2 * S + T;
    const long long bk = blockIdx.x / per_bk;
    const int r = (int)(blockIdx.x - bk * per_bk);
    if (bk >= n_bk) return;
    const double *src = compact + (src_mode == 2 ? bk * CPER : 0);
    double *dst = full + bk * FPER;
    const int tid = threadIdx.x;
    constexpr int Q = (NN / 2 + NT - 1) / NT;
    d2 v[8];
    if (r < 2 * S) {
        const int sign = r / S, sl = r - sign * S;
        const int c0 = sl * cpi, c1 = min(D, c0 + cpi);
        src += sign * NN;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (src_mode == 0)
                v[q] = d2{(double)tid, (double)q};
            else if (tid + NT * q < NN / 2)
                v[q] = *(const d2 *)(src + 2 * (tid + NT * q));
        }
        double *o = dst + ((long long)sign * D + c0) * NN;
        for (int c = c0; c < c1; ++c, o += NN) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (tid + NT * q < NN / 2) {
                    if (sc)
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(o + 2 * (tid + NT * q)), "v"(v[q]) : "memory");
                    else
                        *(d2 *)(o + 2 * (tid + NT * q)) = v[q];
                }
        }
    } else {
        const long long e0 = (long long)(r - 2 * S) * TP;
        src += 2 * NN;
        dst += 2LL * D * NN;
        for (int q = 0; q < TP / NT; ++q) {
            const long long e = e0 + tid + NT * q;
            if (e < tail2) *(d2 *)(dst + 2 * e) = src_mode == 0 ? d2{1.0, 2.0} : *(const d2 *)(src + (src_mode == 2 ? 2 * e : 2 * (e % 1024)));
        }
    }
}

// the same bytes as a plain fill with the expander's grid geometry but NO structure: workgroup w writes bytes [w * chunk, (w + 1) * chunk)
__global__ __launch_bounds__(256) void fill(double *__restrict__ full, long long total2, int pairs_per_wg) {
    const long long e0 = (long long)blockIdx.x * pairs_per_wg;
    for (int q = threadIdx.x; q < pairs_per_wg; q += 256)
        if (e0 + q < total2) *(d2 *)(full + 2 * (e0 + q)) = d2{1.0, (double)q};
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const long long n_bk = 99LL * B;
    double *full, *compact;
    CK(hipMalloc(&full, n_bk * FPER * 8));
    CK(hipMalloc(&compact, n_bk * CPER * 8));
    CK(hipMemset(compact, 0, n_bk * CPER * 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double gb = n_bk * FPER * 8 / 1e9;
    auto timeit = [&](const char *name, auto &&launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipStreamSynchronize(s));
        std::vector<float> ts;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms / 10 * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-64s %8.1f us  %5.2f TB/s\n", name, ts[2], gb / ts[2] * 1e3);
        fflush(stdout);
    };
    timeit("hipMemsetAsync", [&] { CK(hipMemsetAsync(full, 0, n_bk * FPER * 8, s)); });
    for (int ppw : {256, 1024, 4096, 8192})
        timeit((std::string("plain fill, pairs per workgroup ") + std::to_string(ppw)).c_str(), [&] {
            const long long total2 = n_bk * FPER / 2;
            hipLaunchKernelGGL(fill, dim3((unsigned)((total2 + ppw - 1) / ppw)), dim3(256), 0, s, full, total2, ppw);
        });
    for (int src_mode : {0, 1, 2})
        for (int cpi : {1, 3, 9, 27})
            for (int sc : {0, 1}) {
                char nm[128];
                snprintf(nm, sizeof nm, "expand<256> src %s, %2d columns per workgroup%s", src_mode == 0 ? "registers" : src_mode == 1 ? "one tile  " : "compact   ", cpi, sc ? ", sc0 sc1" : "");
                const int S = (D + cpi - 1) / cpi, T = (int)((XD * (M + 1) / 2 + 2047) / 2048);
                timeit(nm, [&] { hipLaunchKernelGGL(expand<256>, dim3((unsigned)(n_bk * (2 * S + T))), dim3(256), 0, s, compact, full, n_bk, cpi, src_mode, sc); });
            }
    for (int cpi : {3, 9}) {
        char nm[128];
        snprintf(nm, sizeof nm, "expand<512> src compact, %2d columns per workgroup", cpi);
        const int S = (D + cpi - 1) / cpi, T = (int)((XD * (M + 1) / 2 + 2047) / 2048);
        timeit(nm, [&] { hipLaunchKernelGGL(expand<512>, dim3((unsigned)(n_bk * (2 * S + T))), dim3(512), 0, s, compact, full, n_bk, cpi, 2, 0); });
    }
    return 0;
}

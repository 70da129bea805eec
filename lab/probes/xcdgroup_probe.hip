// Round 5: how fast is the values array written when ALL workgroups of an XCD write ONE interval's replicated blocks together?
// Bare store pattern (values from registers or from an L2-resident tile), config-3 record geometry (per interval: 27 copies of a 54x54 tile,
// 27 copies of another, a 10,206-double tail).  Persistent workgroups; group g = the workgroups with blockIdx % NG == g (NG = 8: one group per
// XCD under round-robin dispatch) walks the intervals g, g + NG, ...; inside an interval member j writes the block copies j, j + members, ...
// and its share of the tail.  Window of addresses being written = NG intervals (8 x 1.3 MB) instead of 32 x 1.3 MB (round-4 slice tickets)
// or 256 x 0.6 MB (static split).
//   hipcc --offload-arch=gfx950 -O3 -o lab/probes/xcdgroup_probe lab/probes/xcdgroup_probe.hip ; xcdgroup_probe [trajectories=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int D = 27, N = 54, NN = N * N, M = 6;
constexpr long long XD = (long long)N * D, FPER = 2LL * D * NN + XD * (M + 1);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// src 0: registers | 1: the interval's two tiles from a scratch array (loads that bypass the L1: sc1), requested one copy ahead
// piece: a copy (23,328 B) is written by `piece` members together (1: whole copy per member; 2: halves ...) -- finer interleaving of the members
template <int NT>
__global__ __launch_bounds__(NT) void xcdgroup(double *__restrict__ full, const double *__restrict__ scratch, int n_int, int NG, int src, int xcc_out_on, int *xcc_out) {
    const int tid = threadIdx.x, bx = blockIdx.x;
    const int g = bx % NG, j = bx / NG, members = gridDim.x / NG;
    if (xcc_out_on && tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc_out[bx] = (int)(x & 0xf);
    }
    constexpr int Q = (NN / 2 + NT - 1) / NT;
    d2 v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = d2{(double)tid, (double)q};
    const long long tail2 = (XD * (M + 1)) >> 1;
    const int tshare = (int)((tail2 + members - 1) / members);
    for (int iv = g; iv < n_int; iv += NG) {
        double *dst = full + (long long)iv * FPER;
        for (int copy = j; copy < 2 * D; copy += members) {
            if (src == 1) {
                const double *s = scratch + ((long long)iv * 2 + copy / D) * NN;
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    if (tid + NT * q < NN / 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[q]) : "v"(s + 2 * (tid + NT * q)) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            double *o = dst + (long long)copy * NN;
#pragma unroll
            for (int q = 0; q < Q; ++q)
                if (tid + NT * q < NN / 2) *(d2 *)(o + 2 * (tid + NT * q)) = v[q];
        }
        double *t = dst + 2LL * D * NN;
        for (int e = j * tshare + tid; e < min((long long)(j + 1) * tshare, tail2); e += NT) *(d2 *)(t + 2 * e) = d2{1.0, 2.0};
    }
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const int n_int = 99 * B;
    double *full, *scratch;
    int *xcc;
    CK(hipMalloc(&full, (long long)n_int * FPER * 8));
    CK(hipMalloc(&scratch, (long long)n_int * 2 * NN * 8));
    CK(hipMemset(scratch, 0, (long long)n_int * 2 * NN * 8));
    CK(hipMalloc(&xcc, 4096 * sizeof(int)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double gb = (double)n_int * FPER * 8 / 1e9;
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
        printf("%-72s %8.1f us  %5.2f TB/s\n", name, ts[2], gb / ts[2] * 1e3);
        fflush(stdout);
    };
    // the dispatch's XCD of every workgroup of a 256-workgroup launch
    hipLaunchKernelGGL(xcdgroup<256>, dim3(256), dim3(256), 0, s, full, scratch, 8, 8, 0, 1, xcc);
    CK(hipStreamSynchronize(s));
    std::vector<int> hx(256);
    CK(hipMemcpy(hx.data(), xcc, 256 * sizeof(int), hipMemcpyDeviceToHost));
    printf("XCC_ID of workgroups 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hx[i] != hx[i % 8];
    printf("   (workgroups whose XCD differs from that of workgroup bx %% 8: %d of 256)\n", bad);
    timeit("hipMemsetAsync", [&] { CK(hipMemsetAsync(full, 0, (long long)n_int * FPER * 8, s)); });
    for (int src : {0, 1})
        for (int grid : {256, 512})
            for (int NG : {8, 16, 32, 64}) {
                char nm[160];
                snprintf(nm, sizeof nm, "groups %2d x %3d workgroups (grid %d), <256>, src %s", NG, grid / NG, grid, src ? "L2 tile (sc1 loads)" : "registers");
                timeit(nm, [&] { hipLaunchKernelGGL(xcdgroup<256>, dim3(grid), dim3(256), 0, s, full, scratch, n_int, NG, src, 0, xcc); });
            }
    for (int NG : {8, 16}) {
        char nm[160];
        snprintf(nm, sizeof nm, "groups %2d x %3d workgroups (grid 256), <512>, src registers", NG, 256 / NG);
        timeit(nm, [&] { hipLaunchKernelGGL(xcdgroup<512>, dim3(256), dim3(512), 0, s, full, scratch, n_int, NG, 0, 0, xcc); });
    }
    return 0;
}

// Round 5: a grid-strided fill (256 workgroups x 256 threads: every step the grid writes ONE 1 MB window, workgroup w its w-th 4 KB) runs at 6.7 TB/s, the same
// workgroups on contiguous ranges of their own at 5.0-5.5.  Is it the PHASE of the 256 address streams against each other?  Workgroup w owns [w L, (w + 1) L) and
// walks it in 4 KB steps starting `w x skew` bytes into it (wrapping): the streams' addresses modulo a power of two differ by w (L + skew).
//   hipcc --offload-arch=gfx950 -O3 -o lab/probes/phase_probe lab/probes/phase_probe.hip ; phase_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void ranges(char *base, long long L, long long skew, int chunk) {
    const int tid = threadIdx.x;
    const d2 v = d2{(double)tid, 1.0};
    char *lo = base + (long long)blockIdx.x * L;
    const long long steps = L / chunk;
    long long s = ((long long)blockIdx.x * skew / chunk) % steps;
    for (long long i = 0; i < steps; ++i) {
        char *q = lo + s * chunk;
        for (int o = tid * 16; o < chunk; o += 4096) *(d2 *)(q + o) = v;
        if (++s == steps) s = 0;
    }
}

int main() {
    const long long total = 1062LL << 20;
    char *buf;
    CK(hipMalloc(&buf, 256LL * ((total / 256) + (2LL << 20))));  // (every L below fits)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](long long L, long long skew, int chunk) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 12; ++i) {
                if (i == 2) CK(hipEventRecord(e0));
                hipLaunchKernelGGL(ranges, dim3(256), dim3(256), 0, 0, buf, L, skew, chunk);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 10 * 1e3f);
        }
        return best;
    };
    const long long L0 = (total / 256) & ~((1LL << 20) - 1);  // 4 MB
    printf("256 workgroups x 256 threads, ranges of about %lld B each, 4 KB per step; us per launch (TB/s)\n", L0);
    for (long long dL : {0LL, 4096LL, 8192LL, 16384LL, 65536LL, 4096LL * 3, 4096LL * 17, 1LL << 19, 23328LL * 16, 4149248LL - L0})
        for (long long skew : {0LL, 4096LL, 65536LL}) {
            const long long L = L0 + dL;
            const float us = timeit(L, skew, 4096);
            printf("L = 4 MB + %8lld  skew %6lld : %7.1f us (%.2f TB/s)\n", dL, skew, us, 256.0 * L / us / 1e6);
        }
    return 0;
}

// Round 5: how tight must the write front be?  256 workgroups x 256 threads in `ng` groups; group g owns a contiguous ng-th of the array and its members
// walk it grid-strided (every step the group writes ONE window of members x chunk bytes).  ng = 1 is the fill that runs at 6.7 TB/s, ng = 256 the
// ranges of their own (5.0-5.6).  `two`: every chunk is written in two halves 630 KB apart (kernel 4's -B+ / B- segments of a state column).
//   hipcc --offload-arch=gfx950 -O3 -o lab/probes/front_groups_probe lab/probes/front_groups_probe.hip ; front_groups_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void groups(char *base, long long total, int ng, int chunk, int two) {
    const int tid = threadIdx.x, members = gridDim.x / ng, g = blockIdx.x / members, w = blockIdx.x % members;
    const d2 v = d2{(double)tid, 1.0};
    const long long L = total / ng / chunk * chunk;
    char *lo = base + g * L;
    for (long long o = (long long)w * chunk; o + chunk <= L; o += (long long)members * chunk) {
        if (!two) {
            for (int b = tid * 16; b < chunk; b += 4096) *(d2 *)(lo + o + b) = v;
        } else {  // the chunk's halves at o / 2 and L / 2 + o / 2
            for (int b = tid * 16; b < chunk / 2; b += 4096) *(d2 *)(lo + o / 2 + b) = v;
            for (int b = tid * 16; b < chunk / 2; b += 4096) *(d2 *)(lo + L / 2 + o / 2 + b) = v;
        }
    }
}

int main() {
    const long long total = 1062LL << 20;
    char *buf;
    CK(hipMalloc(&buf, total + (16 << 20)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](int ng, int chunk, int two) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 12; ++i) {
                if (i == 2) CK(hipEventRecord(e0));
                hipLaunchKernelGGL(groups, dim3(256), dim3(256), 0, 0, buf, total, ng, chunk, two);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 10 * 1e3f);
        }
        return best;
    };
    printf("1.11 GB, 256 workgroups x 256 threads; us per launch (TB/s)\n");
    for (int two : {0, 1})
        for (int chunk : {4096, 8192, 32768}) {
            printf("chunk %5d B%s:", chunk, two ? ", two halves" : "            ");
            for (int ng : {1, 2, 4, 8, 16, 32, 64, 128, 256}) {
                const float us = timeit(ng, chunk, two);
                printf("  ng %3d: %5.1f (%.2f)", ng, us, total / us / 1e6);
            }
            printf("\n");
        }
    return 0;
}

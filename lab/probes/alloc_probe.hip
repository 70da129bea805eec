// Round 6: WHERE does the "placement lottery" of the multi-trajectory launch come from, and does an allocator remove it?
// (review item: the static split takes 181-228 us per 8 seeds depending on the values array; tickets 193-217 wherever it lives.)
//
// Part A -- allocators.  The static split's store pattern (config-3 record geometry, one persistent workgroup of 4 store waves per CU, equal
// contiguous column ranges: static_variants.hip, mode 0) on arrays from
//   malloc        hipMalloc
//   contiguous    hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   vmm-one       hipMemCreate of the whole array as ONE physical allocation, mapped with hipMemMap
//   vmm-chunks    one hipMemCreate per granule (hipMemGetAllocationGranularity, recommended), mapped in creation order
//   vmm-shuffled  ... the same granules mapped in a random order (the virtual -> physical map is scrambled at granule size)
// `nbuf` arrays of each kind alive at once, every array timed with the pattern and with hipMemsetAsync.
//
// Part B -- which address bits.  256 workgroups, each streaming LEN bytes from base + w * STRIDE: the front of 256 store streams advances in lock
// step, so every moment sees the addresses {base + w STRIDE + t}.  On physically contiguous memory (contiguous / vmm-one) STRIDE decides which
// channels the fronts share; on scattered pages it should not matter.  Sweep STRIDE = LEN + delta.
//
// hipcc --offload-arch=gfx950 -O3 -o lab/probes/alloc_probe lab/probes/alloc_probe.hip ; alloc_probe [nbuf=4] [trajectories=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <string>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int D = 27, N = 54, NN = N * N, HN = 27, M = 6;
constexpr long long BLK = (long long)D * NN, JAC_PER = 2 * BLK + (long long)N * D * (M + 1);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) at line %d\n", hipGetErrorString(e_), (int)e_, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void emu(double *jac, int n_int) {  // the static split (static_variants.hip, mode 0)
    const int tid = threadIdx.x, w = blockIdx.x, G = gridDim.x;
    double bp[6][2], bm[6][2];
#pragma unroll
    for (int r = 0; r < 6; ++r) bp[r][0] = tid + r, bp[r][1] = -tid, bm[r][0] = 0.5 * tid, bm[r][1] = r;
    const long long tot = (long long)n_int * D;
    const long long lo = tot * w / G, hi = tot * (w + 1) / G;
    const int pi = 2 * (tid % HN), pj0 = tid / HN;
    for (long long c = lo; c < hi; ++c) {
        const long long bk = c / D;
        const int cq = (int)(c - bk * D);
        double *o = jac + bk * JAC_PER + (long long)cq * NN;
        if (pj0 < 9) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int j = pj0 + 9 * r;
                *(d2 *)(o + N * j + pi) = d2{bp[r][0], bp[r][1]};
                *(d2 *)(o + BLK + N * j + pi) = d2{bm[r][0], bm[r][1]};
            }
        }
        double *t = jac + bk * JAC_PER + 2 * BLK + (long long)cq * (M + 1) * N;
        for (int e2 = tid; e2 < (M + 1) * HN; e2 += 256) *(d2 *)(t + 2 * e2) = d2{1.0, 2.0};
    }
}
// Part B: workgroup w streams `len` bytes from base + w * stride, 4 KB per step (256 lanes x 16 bytes): 256 fronts in lock step
__global__ __launch_bounds__(256) void fronts(char *base, long long stride, long long len) {
    char *p = base + (long long)blockIdx.x * stride + threadIdx.x * 16;
    const d2 v = d2{1.0 + threadIdx.x, 2.0};
    for (long long o = 0; o < len; o += 4096) *(d2 *)(p + o) = v;
}

// Part C: TWO streams per workgroup, as the static split has them (a column's -B+ copy and its B- copy, BLK * 8 = 629,856 bytes apart): workgroup w
// writes `piece` bytes to base + w * stride + o, then `piece` bytes to the same place + off2, o advancing by `piece`.  (piece = 3888: what one store
// instruction of the fused kernel covers; the pieces are not line-aligned, as in the kernel.)
__global__ __launch_bounds__(256) void fronts2(char *base, long long stride, long long len, long long off2, int piece) {
    char *p = base + (long long)blockIdx.x * stride + threadIdx.x * 16;
    const d2 v = d2{1.0 + threadIdx.x, 2.0};
    const bool on = threadIdx.x * 16 < piece;
    for (long long o = 0; o < len; o += piece)
        if (on) {
            *(d2 *)(p + o) = v;
            *(d2 *)(p + o + off2) = v;
        }
}

struct Arr {
    std::string kind;
    double *p = nullptr;
};
static hipEvent_t e0, e1;
template <class F> static float time_us(F f, int reps) {
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / reps * 1e3f);
    }
    return best;
}
static double *vmm_alloc(size_t bytes, size_t gran, int mode, std::mt19937 &rng) {  // mode 0: one handle | 1: a handle per granule | 2: ... mapped in shuffled order
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t size = (bytes + gran - 1) / gran * gran;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, size, gran, nullptr, 0));
    if (mode == 0) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, size, &prop, 0));
        CK(hipMemMap(va, size, 0, h, 0));
    } else {
        const size_t n = size / gran;
        std::vector<hipMemGenericAllocationHandle_t> hs(n);
        for (auto &h : hs) CK(hipMemCreate(&h, gran, &prop, 0));
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), (size_t)0);
        if (mode == 2) std::shuffle(order.begin(), order.end(), rng);
        for (size_t i = 0; i < n; ++i) CK(hipMemMap((char *)va + i * gran, gran, 0, hs[order[i]], 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, size, &acc, 1));
    return (double *)va;
}

int main(int argc, char **argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 4, ntraj = argc > 2 ? atoi(argv[2]) : 8;
    const int n_int = ntraj * 99, reps = ntraj > 16 ? 4 : 10;
    const size_t bytes = (size_t)n_int * JAC_PER * 8;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("array %.1f MB (%d trajectories); VMM granularity minimum %zu, recommended %zu bytes\n", bytes / 1e6, ntraj, gmin, grec);
    std::mt19937 rng(12345);
    std::vector<Arr> arrs;
    const size_t gran_chunks = std::max<size_t>(grec, (size_t)2 << 20);
    for (int i = 0; i < nbuf; ++i) {
        Arr a;
        a.kind = "malloc";
        CK(hipMalloc(&a.p, bytes));
        arrs.push_back(a);
    }
    for (int i = 0; i < std::min(nbuf, 2); ++i) {
        Arr a;
        a.kind = "contiguous";
        if (hipExtMallocWithFlags((void **)&a.p, bytes, hipDeviceMallocContiguous) != hipSuccess) {
            printf("hipExtMallocWithFlags(contiguous) failed\n");
            (void)hipGetLastError();
            break;
        }
        arrs.push_back(a);
    }
    for (int mode = 0; mode < 3; ++mode)
        for (int i = 0; i < nbuf; ++i) {
            Arr a;
            a.kind = mode == 0 ? "vmm-one" : mode == 1 ? "vmm-chunks" : "vmm-shuffled";
            a.p = vmm_alloc(bytes, mode == 0 ? grec : gran_chunks, mode, rng);
            arrs.push_back(a);
        }
    printf("\nPart A: static-split store pattern / hipMemsetAsync, us per launch (best of 3 x %d)\n", reps);
    std::vector<float> tp(arrs.size()), tm(arrs.size());
    for (int round = 0; round < 2; ++round)
        for (size_t i = 0; i < arrs.size(); ++i) {
            const size_t j = round ? arrs.size() - 1 - i : i;
            double *p = arrs[j].p;
            const float a = time_us([&] { hipLaunchKernelGGL(emu, dim3(256), dim3(256), 0, 0, p, n_int); }, reps);
            const float b = time_us([&] { CK(hipMemsetAsync(p, 0, bytes, 0)); }, reps);
            tp[j] = round ? std::min(tp[j], a) : a;
            tm[j] = round ? std::min(tm[j], b) : b;
        }
    for (size_t i = 0; i < arrs.size(); ++i) printf("%-14s %p  pattern %7.1f  memset %7.1f\n", arrs[i].kind.c_str(), (void *)arrs[i].p, tp[i], tm[i]);
    for (const char *k : {"malloc", "contiguous", "vmm-one", "vmm-chunks", "vmm-shuffled"}) {
        std::vector<float> s;
        for (size_t i = 0; i < arrs.size(); ++i)
            if (arrs[i].kind == k) s.push_back(tp[i]);
        if (s.empty()) continue;
        std::sort(s.begin(), s.end());
        printf("  %-14s pattern min %.1f median %.1f max %.1f\n", k, s.front(), s[s.size() / 2], s.back());
    }

    // Part B: stride sweep of 256 lock-step fronts, 4 MB each, on one array of each kind (needs 256 x (4 MB + delta) <= the array)
    printf("\nPart B: 256 fronts x LEN bytes from base + w * (LEN + delta), us per launch | GB/s\n");
    const long long LEN = (long long)(bytes / 256) / 4096 * 4096 - (1 << 20);  // leave room for the largest delta below
    const long long deltas[] = {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576};
    for (const char *k : {"malloc", "contiguous", "vmm-one", "vmm-shuffled"}) {
        char *base = nullptr;
        for (auto &a : arrs)
            if (a.kind == k && !base) base = (char *)a.p;
        if (!base) continue;
        printf("%-14s LEN %lld:", k, LEN);
        for (long long dlt : deltas) {
            if (256 * (LEN + dlt) > (long long)bytes) continue;
            const float t = time_us([&] { hipLaunchKernelGGL(fronts, dim3(256), dim3(256), 0, 0, base, LEN + dlt, LEN); }, reps);
            printf(" d=%lld: %.1f", dlt, t);
        }
        printf("\n");
    }
    // the same fronts at a length that is a power of two (2 MiB): strides that are multiples of large powers of two
    {
        const long long L2 = 2 << 20;
        if (256 * (L2 + (1 << 20)) <= (long long)bytes + 0) {
            for (const char *k : {"malloc", "contiguous", "vmm-one", "vmm-shuffled"}) {
                char *base = nullptr;
                for (auto &a : arrs)
                    if (a.kind == k && !base) base = (char *)a.p;
                if (!base) continue;
                printf("%-14s LEN 2 MiB:", k);
                for (long long dlt : deltas) {
                    if (256 * (L2 + dlt) > (long long)bytes) continue;
                    const float t = time_us([&] { hipLaunchKernelGGL(fronts, dim3(256), dim3(256), 0, 0, base, L2 + dlt, L2); }, reps);
                    printf(" d=%lld: %.1f", dlt, t);
                }
                printf("\n");
            }
        }
    }
    // Part C: the two-stream pattern; per workgroup 2 x LEN2 bytes; stride between workgroups = what the static split has (array / 256)
    {
        const long long WST = (long long)(bytes / 256) / 16 * 16;  // ~4.15 MB
        const long long offs[] = {629856, 524288, 655360, 1048576, 1572864, 2097152 - 3888 * 16, 2097152};
        printf("\nPart C: two streams per workgroup (piece 3888 B), LEN per stream, second stream at +off2; us per launch\n");
        for (const char *k : {"malloc", "contiguous", "vmm-one", "vmm-shuffled"}) {
            std::vector<char *> bases;
            for (auto &a : arrs)
                if (a.kind == k) bases.push_back((char *)a.p);
            for (size_t bi = 0; bi < bases.size() && bi < 2; ++bi) {
                printf("%-14s #%zu:", k, bi);
                for (long long off2 : offs) {
                    const long long LEN2 = ((WST - off2 > off2 ? off2 : WST - off2) / 3888) * 3888;  // the two streams of a workgroup do not overlap each other or the next workgroup's
                    if (LEN2 <= 0) continue;
                    const float t = time_us([&] { hipLaunchKernelGGL(fronts2, dim3(256), dim3(256), 0, 0, bases[bi], WST, LEN2, off2, 3888); }, reps);
                    printf(" off2=%lld: %.1f us (%.2f TB/s)", off2, t, 2.0 * 256 * LEN2 / t * 1e-6);
                }
                printf("\n");
            }
        }
        // the static split itself with one of its two block streams left out (variant of `emu`)
    }
    return 0;
}

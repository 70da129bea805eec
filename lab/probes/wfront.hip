// Store-pattern emulator of the fused kernel's block stream (round 4): the EXACT store geometry of pcl_fused_sparse_kernel's stream waves
// (records of 167,670 doubles per interval: 27 copies of one 54 x 54 tile, 27 copies of another 629,856 bytes further, a tail run), values
// from registers, no arithmetic -- against the ORDER in which the records are written and the SHAPE of the grid:
//   mode 0  persistent, contiguous column ranges per workgroup (what the library launches for several trajectories)
//   mode 1  persistent, items of `cpi` columns dealt round-robin (item i -> workgroup i mod grid)
//   mode 2  persistent, items of `cpi` columns taken by ticket (one returning atomic per item)
//   mode 3  one short-lived workgroup per item (grid = items), several resident per CU
//   mode 4  as mode 1, the workgroup index permuted so that the workgroups of one XCD (bx mod 8) take neighbouring items
// and against waves per workgroup (4 ... 16: one column per group of four waves at a time) and the store flavour (plain / nt / sc0 sc1).
// Bare fills (hipMemsetAsync, one 16 KB tile per workgroup) on the same buffers beside them.
//   hipcc --offload-arch=gfx950 -O3 -o lab/probes/wfront lab/probes/wfront.hip ;  wfront [buffers] [trajectories]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int D = 27, N = 54, NN = N * N, HN = 27, M = 6;
constexpr long long BLK = (long long)D * NN, JAC_PER = 2 * BLK + (long long)N * D * (M + 1);

template <int NT>
static __device__ __forceinline__ void st2(double *p, double a, double b) {
    d2 v = {a, b};
    if (NT == 0)
        *(d2 *)p = v;
    else if (NT == 1)
        __builtin_nontemporal_store(v, (d2 *)p);
    else
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

struct Prm {
    double *jac;
    int n_int;     // intervals (batch * K)
    int mode, cpi;  // columns per item (modes 1-4)
    int tails;
    unsigned *ticket;
    unsigned *cnt;  // group modes: one slice counter per interval
    int G;          // group modes: workgroups per group
    int gap;  // mode 2: cycles / 64 every wave sleeps between two items (the fused kernel's stream waves fold the next item's powers there)
};

// one column's two blocks by one group of four waves (threads 0..255 of the group), as the library's stream waves store them
template <int NT>
static __device__ __forceinline__ void column_blocks(double *o, int gt, const double (&bp)[6][2], const double (&bm)[6][2]) {
    const int pi = 2 * (gt % HN), pj0 = gt / HN;
    if (pj0 < 9) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int j = pj0 + 9 * r;
            st2<NT>(o + N * j + pi, bp[r][0], bp[r][1]);
            st2<NT>(o + BLK + N * j + pi, bm[r][0], bm[r][1]);
        }
    }
}

template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void emu(const Prm p) {
    const int tid = threadIdx.x, grp = tid >> 8, gt = tid & 255;
    constexpr int NG = NW == 5 ? 1 : NW / 4;
    double bp[6][2], bm[6][2];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        bp[r][0] = tid + r;
        bp[r][1] = -tid;
        bm[r][0] = 0.5 * tid;
        bm[r][1] = r;
    }
    auto do_item = [&](long long col0, int ncols) {  // columns [col0, col0 + ncols) of the global column index (interval * D + c), inside one interval
        const long long bk = col0 / D;
        const int c0 = (int)(col0 - bk * D);
        double *base = p.jac + bk * JAC_PER;
        if (NW == 5 && tid >= 256) return;  // (the fifth wave only takes tickets)
        for (int cq = c0 + grp; cq < c0 + ncols; cq += NG) column_blocks<NT>(base + (long long)cq * NN, gt, bp, bm);
        if (p.tails) {  // the item's share of the tail run (and nothing for delta: another array)
            double *t = base + 2 * BLK + (long long)c0 * (M + 1) * N;
            for (int e2 = tid; e2 < ncols * (M + 1) * HN; e2 += NW == 5 ? 256 : 64 * NW) st2<NT>(t + 2 * e2, 1.0, 2.0);
        }
    };
    const long long tot = (long long)p.n_int * D;
    if (p.mode == 0) {
        const long long lo = tot * blockIdx.x / gridDim.x, hi = tot * (blockIdx.x + 1) / gridDim.x;
        for (long long c = lo; c < hi;) {
            const long long e = std::min(hi, (c / D + 1) * D);
            do_item(c, (int)(e - c));
            c = e;
        }
    } else {
        const int ipi = (D + p.cpi - 1) / p.cpi;  // items per interval
        const long long n_items = (long long)p.n_int * ipi;
        auto run = [&](long long it) {
            const long long bk = it / ipi;
            const int s = (int)(it - bk * ipi);
            const int c0 = s * p.cpi;
            do_item(bk * D + c0, std::min(p.cpi, D - c0));
        };
        if (p.mode == 1) {
            for (long long it = blockIdx.x; it < n_items; it += gridDim.x) run(it);
        } else if (p.mode == 4) {
            const int per = gridDim.x / 8;
            const int wx = (blockIdx.x & 7) * per + (blockIdx.x >> 3);  // XCD x takes items x per .. x per + per - 1 of every round
            for (long long it = wx; it < n_items; it += gridDim.x) run(it);
        } else if (p.mode == 2) {
            __shared__ unsigned nxt;
            for (;;) {
                if (tid == 0) nxt = atomicAdd(p.ticket, 1u);
                __syncthreads();
                const unsigned it = nxt;
                __syncthreads();
                if (it >= n_items) break;
                run(it);
                for (int g = 0; g < p.gap; g += 16) __builtin_amdgcn_s_sleep(16);
            }
        } else if (p.mode == 5) {  // ticket requested one item ahead (its latency hides behind the current item's stores)
            __shared__ unsigned nx2[2];
            if (tid == 0) nx2[0] = atomicAdd(p.ticket, 1u);
            for (int par = 0;; par ^= 1) {
                __syncthreads();
                const unsigned it = nx2[par];
                if (it >= n_items) break;
                if (tid == 0) nx2[par ^ 1] = atomicAdd(p.ticket, 1u);
                run(it);
            }
        } else if (p.mode == 7 || p.mode == 8) {
            // GROUPS: workgroups 8g' .. (one per XCD when G = 8) form a group that walks the intervals g, g + n_groups, ... in a STATIC order; inside
            // an interval its members take slices of cpi columns from the interval's own counter, just in time (mode 7: by a storing thread,
            // which drains its wave's stores; mode 8: by a fifth wave that does not store).  The interval -- hence the powers of G a real
            // workgroup needs -- is known in advance; only the slice is assigned late.
            const int n_groups = gridDim.x / p.G, g = blockIdx.x / p.G;
            __shared__ unsigned sl;
            for (long long bk = g; bk < p.n_int; bk += n_groups) {
                for (;;) {
                    if (tid == (p.mode == 8 ? 256 : 0)) sl = atomicAdd(p.cnt + bk, 1u);
                    __syncthreads();
                    const unsigned s_ = sl;
                    __syncthreads();
                    if (s_ >= (unsigned)ipi) break;
                    run(bk * ipi + s_);
                    for (int g_ = 0; g_ < p.gap; g_ += 16) __builtin_amdgcn_s_sleep(16);
                }
            }
        } else if (p.mode >= 10) {  // tickets taken (mode - 10) items before they are stored (the fused kernel's P wave holds a few)
            const int hold = p.mode - 10;
            __shared__ unsigned ring[8];
            if (tid == 0)
                for (int i = 0; i < hold; ++i) ring[i] = atomicAdd(p.ticket, 1u);
            __syncthreads();
            for (int i = 0;; ++i) {
                const unsigned it = ring[i % hold];
                __syncthreads();
                if (it >= n_items) break;
                if (tid == (NW == 5 ? 256 : 64)) ring[i % hold] = atomicAdd(p.ticket, 1u);  // (NW == 5: a wave that does not store)
                run(it);
                __syncthreads();
            }
        } else {
            run(blockIdx.x);
        }
    }
}

// group mode with TWO-wave workgroups (two per CU: while one is between two slices -- ticket, fold -- the other stores): 128 threads cover
// 4 columns of a block per step, 14 steps per block
__global__ __launch_bounds__(128) void emu2(const Prm p) {
    const int tid = threadIdx.x;
    const int pi = 2 * (tid % HN), pj0 = tid / HN;
    const int ipi = (D + p.cpi - 1) / p.cpi;
    const int n_groups = gridDim.x / p.G, g = blockIdx.x / p.G;
    __shared__ unsigned sl;
    for (long long bk = g; bk < p.n_int; bk += n_groups) {
        for (;;) {
            if (tid == 0) sl = atomicAdd(p.cnt + bk, 1u);
            __syncthreads();
            const unsigned s_ = sl;
            __syncthreads();
            if (s_ >= (unsigned)ipi) break;
            const int c0 = s_ * p.cpi, ncols = std::min(p.cpi, D - c0);
            double *base = p.jac + bk * JAC_PER;
            for (int cq = c0; cq < c0 + ncols; ++cq) {
                double *o = base + (long long)cq * NN;
                if (pj0 < 4)
                    for (int r = 0; r < 14; ++r) {
                        const int j = pj0 + 4 * r;
                        if (j < N) {
                            st2<0>(o + N * j + pi, 1.0 + tid, 2.0);
                            st2<0>(o + BLK + N * j + pi, 3.0, 4.0 + r);
                        }
                    }
            }
            double *t = base + 2 * BLK + (long long)c0 * (M + 1) * N;
            for (int e2 = tid; e2 < ncols * (M + 1) * HN; e2 += 128) st2<0>(t + 2 * e2, 1.0, 2.0);
            for (int g_ = 0; g_ < p.gap; g_ += 16) __builtin_amdgcn_s_sleep(16);
        }
    }
}

template <int U, int NT>
__global__ __launch_bounds__(NT) void fill_tile(d2 *p, size_t n2) {
    const size_t t0 = (size_t)blockIdx.x * NT * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = t0 + (size_t)u * NT + threadIdx.x;
        if (i < n2) p[i] = d2{1.0, 2.0};
    }
}

int main(int argc, char **argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 8;
    const int ntraj = argc > 2 ? atoi(argv[2]) : 8;
    const int reps = ntraj > 16 ? 4 : 10;
    const char *only = argc > 3 ? argv[3] : nullptr;  // comma-separated variant names (default: all)
    auto wanted = [&](const char *name) {
        if (!only) return true;
        const std::string o = std::string(",") + only + ",", n = std::string(",") + name + ",";
        return o.find(n) != std::string::npos;
    };
    const int n_int = ntraj * 99;
    const size_t bytes = (size_t)n_int * JAC_PER * 8;
    std::vector<double *> bufs(nbuf);
    for (auto &b : bufs)
        if (hipMalloc(&b, bytes) != hipSuccess) {
            printf("hipMalloc failed\n");
            return 1;
        }
    unsigned *ticket, *cnt;
    hipMalloc(&ticket, 4);
    hipMalloc(&cnt, 4 * (size_t)n_int);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms / reps * 1000);
        }
        return best;
    };
    struct V {
        const char *name;
        int mode, cpi, grid, nw, nt, tails;
    };
    std::vector<V> vs = {
        {"grp8-3-g128", 7, 3, 128, 4, 108, 1}, {"grp8-3-g160", 7, 3, 160, 4, 108, 1}, {"grp8-3-g192", 7, 3, 192, 4, 108, 1}, {"grp8-3-g224", 7, 3, 224, 4, 108, 1},
        {"grp8-3-g192-gap8k", 7, 3, 192, 4, 108, 1 + 2 * 8}, {"grp8-3-g128-gap8k", 7, 3, 128, 4, 108, 1 + 2 * 8}, {"tick3-g192", 2, 3, 192, 4, 0, 1}, {"tick3-g128", 2, 3, 128, 4, 0, 1},
        {"half16-3", 9, 3, 512, 2, 116, 1}, {"half16-3-gap8k", 9, 3, 512, 2, 116, 1 + 2 * 8}, {"half16-3-gap12k", 9, 3, 512, 2, 116, 1 + 2 * 12}, {"half16-3-gap16k", 9, 3, 512, 2, 116, 1 + 2 * 16},
        {"tick3-gap6k", 2, 3, 256, 4, 0, 1 + 2 * 6}, {"tick3-gap8k", 2, 3, 256, 4, 0, 1 + 2 * 8}, {"tick3-gap12k", 2, 3, 256, 4, 0, 1 + 2 * 12},
        {"grp8-3-gap4k", 7, 3, 256, 4, 108, 1 + 2 * 4}, {"grp8-3-gap8k", 7, 3, 256, 4, 108, 1 + 2 * 8}, {"grp8-3-gap12k", 7, 3, 256, 4, 108, 1 + 2 * 12},
        {"grp8-3", 7, 3, 256, 4, 108, 1}, {"grp8-2", 7, 2, 256, 4, 108, 1}, {"grp8-5", 7, 5, 256, 4, 108, 1}, {"grp4-3", 7, 3, 256, 4, 104, 1}, {"grp16-3", 7, 3, 256, 4, 116, 1}, {"grp32-3", 7, 3, 256, 4, 132, 1},
        {"grp8-3w", 8, 3, 256, 5, 108, 1}, {"grp16-3w", 8, 3, 256, 5, 116, 1}, {"grp8-1w", 8, 1, 256, 5, 108, 1}, {"grp8-1", 7, 1, 256, 4, 108, 1},
        {"tick6-gap1k", 2, 6, 256, 4, 0, 1 + 2 * 1}, {"tick6-gap2k", 2, 6, 256, 4, 0, 1 + 2 * 2}, {"tick6-gap4k", 2, 6, 256, 4, 0, 1 + 2 * 4}, {"tick6-gap8k", 2, 6, 256, 4, 0, 1 + 2 * 8},
        {"tick3-gap1k", 2, 3, 256, 4, 0, 1 + 2 * 1}, {"tick3-gap2k", 2, 3, 256, 4, 0, 1 + 2 * 2}, {"tick3-gap4k", 2, 3, 256, 4, 0, 1 + 2 * 4}, {"tick9-gap4k", 2, 9, 256, 4, 0, 1 + 2 * 4},
        {"tick27-gap4k", 2, 27, 256, 4, 0, 1 + 2 * 4},
        {"hold1-3", 11, 3, 256, 5, 0, 1}, {"hold2-3", 12, 3, 256, 5, 0, 1}, {"hold4-3", 14, 3, 256, 5, 0, 1}, {"hold1-6", 11, 6, 256, 5, 0, 1}, {"hold2-6", 12, 6, 256, 5, 0, 1}, {"hold4-6", 14, 6, 256, 5, 0, 1}, {"hold6-6", 16, 6, 256, 5, 0, 1},
        {"tick2", 2, 2, 256, 4, 0, 1},       {"tick3", 2, 3, 256, 4, 0, 1},       {"tick4", 2, 4, 256, 4, 0, 1},        {"tick5", 2, 5, 256, 4, 0, 1},
        {"tick6", 2, 6, 256, 4, 0, 1},       {"tick9", 2, 9, 256, 4, 0, 1},       {"tick14", 2, 14, 256, 4, 0, 1},      {"tick27", 2, 27, 256, 4, 0, 1},
        {"ptick1", 5, 1, 256, 4, 0, 1},      {"ptick2", 5, 2, 256, 4, 0, 1},      {"ptick3", 5, 3, 256, 4, 0, 1},       {"ptick4", 5, 4, 256, 4, 0, 1},
        {"ptick5", 5, 5, 256, 4, 0, 1},      {"ptick9", 5, 9, 256, 4, 0, 1},      {"ptick3-512", 5, 3, 512, 4, 0, 1},   {"ptick3-8w", 5, 3, 256, 8, 0, 1},
        {"ptick3-wt", 5, 3, 256, 4, 2, 1},   {"ptick3-g198", 5, 3, 198, 4, 0, 1}, {"ptick3-g224", 5, 3, 224, 4, 0, 1},  {"range198", 0, 0, 198, 4, 0, 1},
    };
    std::vector<V> vs_old = {
        {"range256", 0, 0, 256, 4, 0, 1},    {"range256-8w", 0, 0, 256, 8, 0, 1}, {"range256-16w", 0, 0, 256, 16, 0, 1}, {"range512", 0, 0, 512, 4, 0, 1},
        {"rr27", 1, 27, 256, 4, 0, 1},       {"rr9", 1, 9, 256, 4, 0, 1},         {"rr3", 1, 3, 256, 4, 0, 1},          {"rr1", 1, 1, 256, 4, 0, 1},
        {"rr9-512", 1, 9, 512, 4, 0, 1},     {"rr3-512", 1, 3, 512, 4, 0, 1},     {"rr3-1024", 1, 3, 1024, 4, 0, 1},    {"rr9-16w", 1, 9, 256, 16, 0, 1},
        {"rr3-8w", 1, 3, 256, 8, 0, 1},      {"xcd9", 4, 9, 256, 4, 0, 1},        {"xcd3", 4, 3, 256, 4, 0, 1},         {"xcd1", 4, 1, 256, 4, 0, 1},
        {"tick9", 2, 9, 256, 4, 0, 1},       {"tick3", 2, 3, 256, 4, 0, 1},       {"tick1", 2, 1, 256, 4, 0, 1},        {"tick1-512", 2, 1, 512, 4, 0, 1},
        {"short9", 3, 9, 0, 4, 0, 1},        {"short3", 3, 3, 0, 4, 0, 1},        {"short1", 3, 1, 0, 4, 0, 1},         {"rr3-nt", 1, 3, 256, 4, 1, 1},
        {"rr3-wt", 1, 3, 256, 4, 2, 1},      {"range256-notail", 0, 0, 256, 4, 0, 0},
    };
    printf("%d buffers of %.1f MB (%d trajectories); us per launch, best of 3 x %d\n", nbuf, bytes / 1e6, ntraj, reps);
    printf("%-18s", "variant");
    for (int b = 0; b < nbuf; ++b) printf(" buf%-4d", b);
    printf("  median  TB/s(median)\n");
    auto row = [&](const char *name, auto f) {
        if (!wanted(name)) return;
        std::vector<float> t;
        printf("%-18s", name);
        for (int b = 0; b < nbuf; ++b) {
            t.push_back(timeit([&] { f(bufs[b]); }));
            printf(" %7.1f", t.back());
        }
        std::sort(t.begin(), t.end());
        const float med = t[t.size() / 2];
        printf("  %7.1f  %5.2f\n", med, bytes / med / 1e6);
        fflush(stdout);
    };
    row("memset", [&](double *p) { hipMemsetAsync(p, 0, bytes, 0); });
    row("tile16K", [&](double *p) { fill_tile<4, 256><<<(unsigned)((bytes / 16 + 1023) / 1024), 256>>>((d2 *)p, bytes / 16); });
    row("tile8K", [&](double *p) { fill_tile<2, 256><<<(unsigned)((bytes / 16 + 511) / 512), 256>>>((d2 *)p, bytes / 16); });
    row("tile4K", [&](double *p) { fill_tile<1, 256><<<(unsigned)((bytes / 16 + 255) / 256), 256>>>((d2 *)p, bytes / 16); });
    for (const V &v : vs) {
        row(v.name, [&](double *p) {
            Prm prm{p, n_int, v.mode, v.cpi, v.tails & 1, ticket, cnt, v.nt >= 100 ? v.nt - 100 : 8, (v.tails >> 1) * 16};
            if (v.mode == 7 || v.mode == 8 || v.mode == 9) hipMemsetAsync(cnt, 0, 4 * (size_t)n_int, 0);
            int grid = v.grid;
            if (v.mode == 3) grid = n_int * ((D + v.cpi - 1) / v.cpi);
            if (v.mode == 2 || v.mode == 5 || v.mode >= 10) hipMemsetAsync(ticket, 0, 4, 0);
#define L(NT_, NW_) emu<NT_, NW_><<<grid, 64 * NW_>>>(prm)
            if (v.mode == 9)
                emu2<<<grid, 128>>>(prm);
            else if (v.mode == 8)
                L(0, 5);
            else if (v.mode == 7)
                L(0, 4);
            else if (v.nt == 2 && v.mode == 5)
                L(2, 4);
            else if (v.nt == 1)
                L(1, 4);
            else if (v.nt == 2)
                L(2, 4);
            else if (v.nw == 5)
                L(0, 5);
            else if (v.nw == 8)
                L(0, 8);
            else if (v.nw == 16)
                L(0, 16);
            else
                L(0, 4);
        });
    }
    return 0;
}

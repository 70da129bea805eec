// Micro-benchmark of the generated product G(u)^T x (config 3, order 8 module: gen_cfg3_q4.inc = the generated functions of
// pcl_codegen_source_v4(what = 1) without their includes): cycles per call for ONE wave per CU / per SIMD, 1 .. 4 waves per SIMD.
// python make_inc.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o prodbench prodbench.hip && ./prodbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#ifndef PRODBENCH_INC
#define PRODBENCH_INC "gen_cfg3_q4.inc"  // (make_inc.py writes it; -DPRODBENCH_INC='"gen_noadd.inc"': without the products' ds_add_f64)
#endif
#include PRODBENCH_INC
#define CS (SPN + 1)
template <int MODE>
__global__ void bench(const double *mags_, const double *dcf_tab, const double *u_, long long *out, int reps) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    sp_cptr magc = (sp_cptr)mags_;
    double mg[SP4NMAG];
    for (int g = 0; g < SP4NMAG; ++g) mg[g] = magc[g];
    sp_cptr uc = (sp_cptr)u_;
    double u[SPM];
    for (int l = 0; l < SPM; ++l) u[l] = uc[l];
    sp4_cf cf;
    SP4_SET_CF(cf, u, mg);
    SP4_SET_DCF(cf, (sp_cptr)dcf_tab);
    const int half = lane >> 5, c = lane & 31;
    double *tile = lds + wave * 32 * CS;
    const int own = c * CS + half * SPD, oth = c * CS + (1 - half) * SPD;
    for (int i = 0; i < SPD; ++i) tile[own + i] = 1e-3 * (lane + i);
    double x[SPD];
    for (int i = 0; i < SPD; ++i) x[i] = 1e-3 * (i + lane);
    const unsigned oX = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)(tile + own);
    const unsigned oXx = (unsigned)(size_t)(__attribute__((address_space(3))) const double *)(tile + oth);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) sp4_product_t(x, oX, oX, oXx, 1.0, 1.0, half ? 1.0 : -1.0, (sp_cptr)dcf_tab, cf);
        if (MODE == 1) sp4_product0_t(x, 0u, oX, oXx, 0.0, 1.0, half ? 1.0 : -1.0, (sp_cptr)dcf_tab, cf);
        if (MODE == 2) sp4_product0(x, 0u, oX, oXx, 0.0, 1.0, half ? -1.0 : 1.0, (sp_cptr)dcf_tab, cf);
        if (MODE == 3) {  // read back (a chain)
            sp4_product_t(x, oX, oX, oXx, 1.0, 1.0, half ? 1.0 : -1.0, (sp_cptr)dcf_tab, cf);
            for (int i = 0; i < SPD; ++i) x[i] = tile[own + i] * 1e-3;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (x[0] == 123.456) out[0] = 0;
}
int main() {
    double hm[8] = {0.3, 0.45, 0, 0, 0, 0, 0, 0}, hd[32], hu[8] = {0.01, -0.02, 0.03, 0.01, 0.02, -0.01, 0, 0};
    for (int i = 0; i < 32; ++i) hd[i] = 0.1 + 0.01 * i;
    double *dm, *dd, *du;
    long long *dout;
    hipMalloc(&dm, sizeof hm); hipMalloc(&dd, sizeof hd); hipMalloc(&du, sizeof hu); hipMalloc(&dout, 256 * 16 * 8);
    hipMemcpy(dm, hm, sizeof hm, hipMemcpyHostToDevice); hipMemcpy(dd, hd, sizeof hd, hipMemcpyHostToDevice); hipMemcpy(du, hu, sizeof hu, hipMemcpyHostToDevice);
    const int reps = 200;
    for (int mode = 0; mode < 4; ++mode)
        for (int waves : {1, 4, 8, 11}) {
            hipMemset(dout, 0, 256 * 16 * 8);
            const size_t lds = (size_t)waves * 32 * CS * 8;
            auto k = mode == 0 ? bench<0> : mode == 1 ? bench<1> : mode == 2 ? bench<2> : bench<3>;
            hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k, dim3(8), dim3(64 * waves), lds, 0, dm, dd, du, dout, reps);
            hipDeviceSynchronize();
            std::vector<long long> h(256 * 16);
            hipMemcpy(h.data(), dout, 8 * 16 * 8, hipMemcpyDeviceToHost);
            long long mx = 0;
            for (int w = 0; w < waves; ++w) mx = std::max(mx, h[w]);
            printf("mode %d (%s) waves/CU %2d: %lld cycles per call (wave 0: %lld)\n", mode, mode == 0 ? "product_t with Y" : mode == 1 ? "product0_t" : mode == 2 ? "product0" : "product_t + read back", waves,
                   mx / reps, h[0] / reps);
        }
    return 0;
}

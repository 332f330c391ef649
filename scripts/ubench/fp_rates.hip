// Micro-benchmark: issue rate of the floating-point vector instructions the softmax / GELU / epilogue code is made of (gfx950):
// plain and packed fp32 arithmetic, the transcendentals (v_exp_f32, v_rcp_f32 and their f16 forms), conversions.
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/fp_rates.bin scripts/ubench/fp_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a = threadIdx.x * 1e-3f + seed, b = a + 0.25f, c = b * 0.5f, d = c + 0.125f, e = 0.999f, f = 1e-3f;
    f2 pa = {a, b}, pb = {c, d}, pc = {e, f}, pd = {b, c}, pe = {0.999f, 1.001f}, pf = {1e-3f, 2e-3f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pe), "v"(pf));
            if (OP == 2) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 3) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 4) asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 5) asm volatile("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pe));
            if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pf));
            if (OP == 8) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_pk_f16_f32 %3, %1, %2\n v_cvt_pk_f16_f32 %4, %1, %2\n v_cvt_pk_f16_f32 %5, %1, %2" : "=v"(a), "+v"(e), "+v"(f), "=v"(b), "=v"(c), "=v"(d));
            if (OP == 9) asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 10) asm volatile("v_cvt_f32_f16 %0, %1\n v_cvt_f32_f16 %3, %2\n v_cvt_f32_f16 %4, %1\n v_cvt_f32_f16 %5, %2" : "=v"(a), "+v"(e), "+v"(f), "=v"(b), "=v"(c), "=v"(d));
            if (OP == 11) asm volatile("v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 12) asm volatile("v_fma_mix_f32 %0, %0, %4, %5\n v_fma_mix_f32 %1, %1, %4, %5\n v_fma_mix_f32 %2, %2, %4, %5\n v_fma_mix_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
            if (OP == 13) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + pa[0] + pa[1] + pb[0] + pb[1] + pc[0] + pc[1] + pd[0] + pd[1];
}
template <int OP> void run(const char* name, int wgs_per_cu) {
    float* out; hipMalloc(&out, 256 * 256 * 16 * 4);
    const int iters = 1000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 256>>>(out, 10, 1.f); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<grid, 256>>>(out, iters, 1.f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)wgs_per_cu * iters * REP * 4;      // one wave per SIMD and workgroup
    printf("%-34s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wgs_per_cu, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<6>("v_pk_mul_f32", w); run<7>("v_pk_add_f32", w); run<12>("v_fma_mix_f32", w); run<11>("v_pk_fma_f16", w);
        run<2>("v_exp_f32", w); run<3>("v_rcp_f32", w); run<4>("v_exp_f16", w); run<5>("v_rcp_f16", w);
        run<8>("v_cvt_pk_f16_f32", w); run<10>("v_cvt_f32_f16", w); run<9>("v_max3_f32", w); run<13>("1 v_exp_f32 + 3 v_fma_f32", w);
    }
    return 0;
}

// Micro-benchmark: issue rate of the integer vector instructions the resize kernels are made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u, c = b * 3u, d = c + 7u;
    uint32_t e = a + 11u, f = b + 13u, g = c + 17u, h = d + 19u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) { asm volatile("v_mad_i32_i24 %0, %1, %2, %0\n v_mad_i32_i24 %3, %4, %5, %3\n v_mad_i32_i24 %6, %7, %8, %6\n v_mad_i32_i24 %9, %10, %11, %9"
                                        : "+v"(a), "+v"(e), "+v"(f), "+v"(b), "+v"(g), "+v"(h), "+v"(c), "+v"(e), "+v"(g), "+v"(d), "+v"(f), "+v"(h)); }
            if (OP == 1) { asm volatile("v_mul_i32_i24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                                        "v_mul_i32_i24_sdwa %3, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
                                        "v_mul_i32_i24_sdwa %4, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
                                        "v_mul_i32_i24_sdwa %5, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                                        : "=v"(a), "+v"(e), "+v"(f), "=v"(b), "=v"(c), "=v"(d)); }
            if (OP == 2) { asm volatile("v_dot4_u32_u8 %0, %1, %2, %0\n v_dot4_u32_u8 %3, %1, %2, %3\n v_dot4_u32_u8 %4, %1, %2, %4\n v_dot4_u32_u8 %5, %1, %2, %5"
                                        : "+v"(a), "+v"(e), "+v"(f), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 3) { asm volatile("v_perm_b32 %0, %1, %2, %6\n v_perm_b32 %3, %1, %2, %6\n v_perm_b32 %4, %1, %2, %6\n v_perm_b32 %5, %1, %2, %6"
                                        : "=v"(a), "+v"(e), "+v"(f), "=v"(b), "=v"(c), "=v"(d) : "v"(g)); }
            if (OP == 4) { asm volatile("v_add3_u32 %0, %1, %2, %0\n v_add3_u32 %3, %1, %2, %3\n v_add3_u32 %4, %1, %2, %4\n v_add3_u32 %5, %1, %2, %5"
                                        : "+v"(a), "+v"(e), "+v"(f), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 5) { asm volatile("v_alignbyte_b32 %0, %1, %2, %6\n v_alignbyte_b32 %3, %1, %2, %6\n v_alignbyte_b32 %4, %1, %2, %6\n v_alignbyte_b32 %5, %1, %2, %6"
                                        : "=v"(a), "+v"(e), "+v"(f), "=v"(b), "=v"(c), "=v"(d) : "v"(g)); }
            if (OP == 6) { asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %1, %2, %3\n v_fma_f32 %4, %1, %2, %4\n v_fma_f32 %5, %1, %2, %5"
                                        : "+v"(a), "+v"(e), "+v"(f), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 7) { asm volatile("v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %3, %1, 16, 8\n v_bfe_u32 %4, %2, 8, 8\n v_bfe_u32 %5, %2, 16, 8"
                                        : "=v"(a), "+v"(e), "+v"(f), "=v"(b), "=v"(c), "=v"(d)); }
            if (OP == 8) { asm volatile("v_mul_u32_u24 %0, %1, %2\n v_mul_u32_u24 %3, %1, %2\n v_mul_u32_u24 %4, %1, %2\n v_mul_u32_u24 %5, %1, %2"
                                        : "=v"(a), "+v"(e), "+v"(f), "=v"(b), "=v"(c), "=v"(d)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <int OP> void run(const char* name, int wgs_per_cu) {
    uint32_t* out; hipMalloc(&out, 256 * 256 * 16 * 4);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, 256>>>(out, 10, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<grid, 256>>>(out, iters, 1); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: waves/SIMD = wgs_per_cu (4 waves per WG, one per SIMD) ; each wave iters*REP*4 instrs
    const double inst_per_simd = (double)wgs_per_cu * iters * REP * 4;
    printf("%-22s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wgs_per_cu, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<6>("v_fma_f32", w); run<0>("v_mad_i32_i24", w); run<1>("v_mul_i32_i24_sdwa", w); run<8>("v_mul_u32_u24", w); run<2>("v_dot4_u32_u8", w);
        run<3>("v_perm_b32", w); run<4>("v_add3_u32", w); run<5>("v_alignbyte_b32", w); run<7>("v_bfe_u32", w);
    }
    return 0;
}

// 256x256x64 bf16/fp16 TN GEMM, fourth-generation mainloop: FOUR waves (one per SIMD, 128 x 128 of C each) whose K loop is
// ONE inline-asm text with fixed registers, scheduled instruction by instruction by scripts/gen_gemm_asm.py
// (gemm_asm256_loop.inc) instead of by the compiler.
//
// Why: every HIP-source mainloop of rounds 1-3 (four-phase, ring, deep prefetch, two per CU, four waves with per-MFMA asm)
// tops out at 1.03-1.06 PFLOP/s on 8192^3 while the MFMA pipe of a SIMD is idle a quarter of the K loop: the compiler
// neither interleaves the fragment reads and the LDS-DMA issues with the MFMAs one by one nor lets barriers sit INSIDE an
// MFMA stream, and with two waves per SIMD the hand-over between them costs a barrier per 16 MFMAs.  With one wave per SIMD
// a 16-cycle MFMA leaves room for one or two other instructions behind it; a wave that issues 128 MFMAs per K-tile back to
// back with its 32 ds_read_b128, 16 buffer_load ... lds and two s_barrier placed between them keeps its pipe busy without a
// partner.  LDS reads drop to 128 KiB per K-tile (8 waves of 128 x 64: 192 KiB).
//
// LDS image, swizzle and DMA pieces: gemm_mfma256.h (row r of A at r*128 B, W at 32 KiB + r*128 B, two 64-KiB K-tile
// buffers); wave w brings rows 64w..64w+63 of both operands.  The schedule and the fixed-register map are documented in
// scripts/gen_gemm_asm.py.  The C++ around the asm issues the first two K-tiles (so that the LayerNorm row statistics of
// the LN-consuming epilogues are computed under them), hands the operands over, and afterwards pulls the accumulators out
// of a[0:255] pass by pass into the strip-transposing epilogue the other kernels use (wave_epilogue_agpr).
//
// The compiler is told that the asm clobbers a0-a255, v116-v255 and s40-s63; it is NOT told that a0-a255 stay live until
// the VQ_A256_READ_HALF reads behind the loop (binding them as "={a[0:15]}" outputs crashes this compiler's backend).
// Nothing between the loop and the read of an accumulator may therefore write its AGPR.  The epilogue (wave_epilogue_agpr
// below) pulls one pass of 8 tiles at a time, and the reads carry a "memory" clobber so that they stay behind the previous
// pass's stores: 32 accumulators are live in VGPRs at a time, every epilogue then needs < 256 VGPRs and the allocator has no
// reason to touch the AGPR half of the file.  scripts/check_asm256.py disassembles every instantiation and fails on any AGPR
// the compiler writes before the asm statement that reads it out.
//
// Requirements: M % 256 == 0, N % 256 == 0, K % 128 == 0, lda / ldw % 8 == 0, 16-byte aligned bases, rows * ld * 2 < 2^31.
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "gemm_mfma256d.h"
#include "gemm_asm256_loop.inc"
#ifdef VQ_DIAG
#include "gemm_asm256_loop_diag.inc"      // generated into $(OBJDIR) by `make DIAG=1` (scripts/gen_gemm_asm.py --diag)
#define VQ_A256_MORE_CLOBBERS VQ_A256_DIAG_CLOBBERS
#else
#define VQ_A256_MORE_CLOBBERS
#endif

namespace vq {

constexpr int GA_THREADS = 256;
constexpr int GA_LDS_BYTES = G2_LDS_BYTES + G2_ROWSTAT_BYTES;      // two K-tile buffers + (mean, rstd) per tile row

// The wave's 128 x 128 accumulators leave a[0:255] one pass (32 rows x 64 columns = 8 MFMA tiles, 32 registers) at a time:
// wave_epilogue<8> of gemm_mfma.h over both column halves in ONE software pipeline - the memory-reading epilogues keep the
// next pass's reads in flight across the half boundary, and only one pass of accumulators is live in VGPRs.  (Two calls of
// wave_epilogue<8> on 128 registers each pushed the residual epilogues past 256 VGPRs: the allocator then spills into the
// AGPR half, over accumulators it does not know are live; four calls of wave_epilogue<4> stayed below but exposed four
// read round trips per wave instead of one: 100.4k frames/s against 108k.)
template <class Epi>
__device__ __forceinline__ void wave_epilogue_agpr(char* strip, int m_wave0, int n_wave0, int lane, const Epi& epi) {
    const int frow = lane & 15, fgrp = lane >> 4;
    if constexpr (epi_wide<Epi>::value) {
        static_assert(epi_row_in<Epi>::value && !Epi::kLoads, "the 8-column epilogue form is for the LayerNorm-consuming 16-bit outputs");
        const int wrow = lane >> 3, wcol = lane & 7;
        f32x4 b0[2], b1[2], a0[2], a1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n8 = n_wave0 + h * 64 + wcol * 8;
            b0[h] = epi.bias_at(n8); b1[h] = epi.bias_at(n8 + 4); a0[h] = epi.aux_at(n8); a1[h] = epi.aux_at(n8 + 4);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n8 = n_wave0 + h * 64 + wcol * 8;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        f32x4 t;
                        VQ_A256_READ_TILE(pass * 2 + hb, h * 4 + ni, t);
                        *(f32x4*)(strip + (hb * 16 + frow) * EPI_ROW_BYTES + (ni * 16 + fgrp * 4) * 4) = t;
                    }
                // same wave, in-order LDS: the reads below see the writes above
                int me = m_wave0 + pass * 32 + wrow;
                asm volatile("" : "+v"(me));              // opaque: the addresses of a pass are computed in that pass, not hoisted for all eight
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + wrow;
                    const f32x4 v0 = *(const f32x4*)(strip + row * EPI_ROW_BYTES + wcol * 32);
                    const f32x4 v1 = *(const f32x4*)(strip + row * EPI_ROW_BYTES + wcol * 32 + 16);
                    const int m = me + it * 8;
                    epi.store_ln8(m, n8, v0, v1, b0[h], b1[h], a0[h], a1[h], epi.row_stat(m));
                }
            }
        }
        return;
    } else {
    const int rrow = lane >> 4, rcol = lane & 15;
    f32x4 bias[2], aux[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n_wave0 + h * 64 + rcol * 4;
        bias[h] = epi.bias_at(n);
        aux[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (epi_row_in<Epi>::value) aux[h] = epi.aux_at(n);
    }
    f32x4 loaded[2][8];
    if constexpr (Epi::kLoads) {
#pragma unroll
        for (int it = 0; it < 8; ++it) loaded[0][it] = epi.load(m_wave0 + it * 4 + rrow, n_wave0 + rcol * 4);
    }
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {              // pass idx & 3 of column half idx >> 2
        const int h = idx >> 2, pass = idx & 3;
        const int n = n_wave0 + h * 64 + rcol * 4;
        if constexpr (Epi::kLoads) {
            if (idx + 1 < 8) {
                const int h1 = (idx + 1) >> 2, p1 = (idx + 1) & 3;
                int m1 = m_wave0 + p1 * 32 + rrow;
                asm volatile("" : "+v"(m1));              // opaque: the addresses of a pass are computed in that pass, not hoisted for all eight
#pragma unroll
                for (int it = 0; it < 8; ++it) loaded[(idx + 1) & 1][it] = epi.load(m1 + it * 4, n_wave0 + h1 * 64 + rcol * 4);
            }
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                f32x4 t;
                VQ_A256_READ_TILE(pass * 2 + hb, h * 4 + ni, t);
                *(f32x4*)(strip + (hb * 16 + frow) * EPI_ROW_BYTES + (ni * 16 + fgrp * 4) * 4) = t;
            }
        // same wave, in-order LDS: the reads below see the writes above
        int me = m_wave0 + pass * 32 + rrow;
        asm volatile("" : "+v"(me));
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const f32x4 v = *(const f32x4*)(strip + (it * 4 + rrow) * EPI_ROW_BYTES + rcol * 16);
            epi_emit(epi, me + it * 4, n, n_wave0 + h * 64, rcol, v, bias[h], aux[h], Epi::kLoads ? loaded[idx & 1][it] : f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
    }
}

template <bool IS_F16, class Epi, int V = 0 /* diagnostic builds: 1-3 = timing ablations of the K loop (gemm_asm256_loop.inc) */>
__global__ __launch_bounds__(GA_THREADS, 1) __attribute__((amdgpu_num_vgpr(256)))
void gemm_tn256a_kernel(const uint16_t* __restrict__ A, int lda,
                        const uint16_t* __restrict__ W, int ldw,
                        int K, int tiles_n, Epi epi, int order2d,
                        unsigned long long* __restrict__ clock_out = nullptr /* diagnostic builds: per workgroup {d s_memtime, d s_memrealtime} around the K loop */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int tm = wg / tiles_n, tn = wg % tiles_n;
    if (order2d) tile_coords(wg, (int)gridDim.x / tiles_n, tiles_n, tm, tn);
    const int m0 = tm * G2_BM;
    const int n0 = tn * G2_BN;

    // ---- LDS-DMA: wave w fills pieces q = 0..7 (8 rows x 128 B) of rows 64w.. of A and of W.  Row 64w + 8q + srow has
    //      swizzle ((row >> 1) & 7) = (srow >> 1) + 4 (q & 1): even pieces use lane offset v0, odd pieces v0 ^ 64. ----
    const int srow = lane >> 3, sslot = lane & 7;
    const int row_w = wave * 64 + srow;
    const int a_v0 = (row_w * lda + (sslot ^ (srow >> 1)) * 8) * 2;
    const int w_v0 = (row_w * ldw + (sslot ^ (srow >> 1)) * 8) * 2;
    const int a_row8 = 8 * lda * 2, w_row8 = 8 * ldw * 2;
    const uint16_t* a_tile = A + (size_t)m0 * lda;
    const uint16_t* w_tile = W + (size_t)n0 * ldw;
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)a_tile, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)w_tile, 0, 0x7fffffff, 0x00020000);
    const int nk = K / G2_BK;

    // tiles 0 and 1 -> buffers 0 and 1, in the order the asm's counted waits assume: per tile 8 A pieces, then 8 W pieces
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        char* base = smem + t * G2_BUF + wave * 8192;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + q * 1024), 16, (q & 1) ? (a_v0 ^ 64) : a_v0,
                                                     q * a_row8 + t * (G2_BK * 2), 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + 2 * G2_HALF + q * 1024), 16, (q & 1) ? (w_v0 ^ 64) : w_v0,
                                                     q * w_row8 + t * (G2_BK * 2), 0, 0);
    }
    // (mean, rstd) of the tile's rows for LayerNorm-consuming epilogues, while the first tiles are in flight
    const Epi epi_wg = epi_bind_rowstats<G2_BM>(epi, (float2*)(smem + G2_LDS_BYTES), m0, tid, GA_THREADS);

    // ---- fragment read addresses (k-half 0; the asm derives k-half 1 = ^64 and buffer 1 = +64 KiB) ----
    const int frow = lane & 15, fgrp = lane >> 4;
    const int fx = (frow >> 1) & 7;
    const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int rd_a = lds0 + wr * G2_HALF + frow * 128 + ((fgrp ^ fx) * 16);
    const int rd_w = lds0 + 2 * G2_HALF + wc * G2_HALF + frow * 128 + ((fgrp ^ fx) * 16);

    {
        const uint64_t pa = (uint64_t)(uintptr_t)a_tile + 2 * (G2_BK * 2), pw = (uint64_t)(uintptr_t)w_tile + 2 * (G2_BK * 2);   // tile 2 onwards
        const int srd_a0 = __builtin_amdgcn_readfirstlane((int)(uint32_t)pa), srd_a1 = __builtin_amdgcn_readfirstlane((int)(uint32_t)(pa >> 32) & 0xffff);
        const int srd_w0 = __builtin_amdgcn_readfirstlane((int)(uint32_t)pw), srd_w1 = __builtin_amdgcn_readfirstlane((int)(uint32_t)(pw >> 32) & 0xffff);
        const int m0_a = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);
        const int trips = __builtin_amdgcn_readfirstlane((nk - 2) >> 1);
        const int a_row8_s = __builtin_amdgcn_readfirstlane(a_row8), w_row8_s = __builtin_amdgcn_readfirstlane(w_row8);
#ifdef VQ_DIAG
        unsigned long long c0 = 0, r0 = 0;
        if (clock_out) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
#define VQ_A256_RUN(TEXT)                                                                                                     \
        asm volatile(TEXT                                                                                                     \
                     :                                                                                                        \
                     : [srd_a0] "s"(srd_a0), [srd_a1] "s"(srd_a1), [srd_w0] "s"(srd_w0), [srd_w1] "s"(srd_w1),                \
                       [a_v0] "v"(a_v0), [w_v0] "v"(w_v0), [rd_a] "v"(rd_a), [rd_w] "v"(rd_w),                                \
                       [m0_a] "s"(m0_a), [trips] "s"(trips), [a_row8] "s"(a_row8_s), [w_row8] "s"(w_row8_s), [wave] "s"(wave)  \
                     : "memory", "scc", VQ_A256_CLOBBERS VQ_A256_MORE_CLOBBERS)
#define VQ_A256_RUN_TY(NAME) do { if constexpr (IS_F16) VQ_A256_RUN(NAME("f16")); else VQ_A256_RUN(NAME("bf16")); } while (0)
#ifdef VQ_DIAG
        if constexpr (V == 1) VQ_A256_RUN_TY(VQ_A256_LOOP_TEXT_1);
        else if constexpr (V == 2) VQ_A256_RUN_TY(VQ_A256_LOOP_TEXT_2);
        else if constexpr (V == 3) VQ_A256_RUN_TY(VQ_A256_LOOP_TEXT_3);
        else if constexpr (V == 4) VQ_A256_RUN_TY(VQ_A256_LOOP_TEXT_4);
        else if constexpr (V == 5) VQ_A256_RUN_TY(VQ_A256_LOOP_TEXT_5);
        else if constexpr (V == 6) VQ_A256_RUN_TY(VQ_A256_LOOP_TEXT_6);
        else if constexpr (V == 7) VQ_A256_RUN_TY(VQ_A256_LOOP_TEXT_7);
        else
#endif
        VQ_A256_RUN_TY(VQ_A256_LOOP_TEXT_0);
#undef VQ_A256_RUN_TY
#undef VQ_A256_RUN
#ifdef VQ_DIAG
        if (clock_out) {
            const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) { clock_out[blockIdx.x * 2] = c1 - c0; clock_out[blockIdx.x * 2 + 1] = r1 - r0; }
        }
#endif
    }
    __syncthreads();                      // every wave is past its last fragment read: the buffers become the epilogue strips

    char* strip = smem + wave * EPI_WAVE_BYTES;
    wave_epilogue_agpr(strip, m0 + wr * 128, n0 + wc * 128, lane, epi_wg);
}

template <bool IS_F16, class Epi, int V = 0>
static int launch_gemm_tn256a(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                              int M, int N, int K, const Epi& epi, unsigned long long* clock_out = nullptr) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0,
             "gemm_tn256a: shape M=%d N=%d K=%d is not tile-aligned (256/256/128)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
             (int64_t)256 * lda * 2 < ((int64_t)1 << 31) && (int64_t)256 * ldw * 2 < ((int64_t)1 << 31),
             "gemm_tn256a: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;       // per instantiation; one device per process (vq_init)
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256a_kernel<IS_F16, Epi, V>, hipFuncAttributeMaxDynamicSharedMemorySize, GA_LDS_BYTES));
        attr_set = true;
    }
    const int tiles_m = M / G2_BM, tiles_n = N / G2_BN;
    hipLaunchKernelGGL((gemm_tn256a_kernel<IS_F16, Epi, V>), dim3(tiles_m * tiles_n), dim3(GA_THREADS), GA_LDS_BYTES, st,
                       A, lda, W, ldw, K, tiles_n, epi, gemm_order2d(), clock_out);
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace vq

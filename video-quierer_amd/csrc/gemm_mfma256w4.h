// C[M,N] = A[M,K] * W[N,K]^T, 256x256 output tile on FOUR waves (one per SIMD), each owning 128x128 of it.
//
// Why a second 256x256 mainloop: removing parts of the eight-wave kernels (scripts/gemm_ablate.py, 16384x4096x4096,
// every CU busy) gives 1668 TFLOP/s for MFMAs + barriers alone, 1246 with the LDS fragment reads added back and
// 1000-1030 complete — the MFMAs wait on fragment reads and on LDS-DMA issue.  Here a wave holds 128x128
// accumulators (256 registers; the 512-entry file is not shared with a second wave), so one 32-wide k-step is
// 64 MFMAs fed by 16 fragment reads (0.25 reads per MFMA instead of 0.375), and the fragments of step p+1 are
// read into a second register set while the 64 MFMAs of step p run: no MFMA ever waits on LDS.  One barrier per
// k-step hands slots over; LDS-DMA runs NSLOT k-steps ahead.
//
// LDS image, swizzle and DMA pieces are those of gemm_tn256_ring_kernel (gemm_mfma256.h): a slot is one 32-wide
// sub-tile = 256 A rows + 256 W rows of 64 B; a 1-KiB piece = 16 rows; waves 0-1 bring the A pieces, waves 2-3 the
// W pieces (8 each per sub-tile).
#pragma once
#include "gemm_mfma256.h"
#include <type_traits>

namespace vq {

constexpr int G4_THREADS = 256;
constexpr int G4_NSLOT = 4;                                  // 4 x 32 KiB; the epilogue strips reuse the ring
constexpr int G4_LDS_BYTES = G4_NSLOT * G3_SLOT;

// MFMA with the accumulator pinned to the AGPR half of the register file ("+a"), in place.  Left to itself the
// register allocator spreads 256 accumulator registers over both halves and shuffles them every k-step
// (800 v_accvgpr moves and scratch traffic in the loop); pinned, the VGPR half is left to the two fragment sets.
typedef int g4_v4i __attribute__((ext_vector_type(4)));
template <bool IS_F16, class Frag>
__device__ __forceinline__ void mfma_agpr(f32x4& c, const Frag& a, const Frag& b) {
    const g4_v4i av = __builtin_bit_cast(g4_v4i, a), bv = __builtin_bit_cast(g4_v4i, b);
    // "memory": the fragment reads and LDS-DMA issues written between two MFMAs are to stay between them
    if constexpr (IS_F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv) : "memory");
    else                  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv) : "memory");
}

template <bool IS_F16, class Epi>
__global__ __launch_bounds__(G4_THREADS, 1)
void gemm_tn256w4_kernel(const uint16_t* __restrict__ A, int lda,
                         const uint16_t* __restrict__ W, int ldw,
                         int K, int tiles_n, Epi epi) {
    typedef typename mfma_op<IS_F16>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * G2_BM;
    const int n0 = (wg % tiles_n) * G2_BN;

    // ---- LDS-DMA sources: wave w < 2 fills A pieces 8w..8w+7, wave w >= 2 fills W pieces 8(w-2).. ----
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ (((lane >> 5) & 1) * 2);
    const bool a_side = wave < 2;
    const uint16_t* gbase = a_side ? A + (size_t)m0 * lda : W + (size_t)n0 * ldw;
    const int gld = a_side ? lda : ldw;
    const int piece0 = (wave & 1) * 8;
    const uint16_t* src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) src[i] = gbase + (size_t)((piece0 + i) * 16 + srow) * gld + schunk * 8;
    const int dst_off = (a_side ? 0 : G3_PART) + piece0 * 1024;

    auto stage_piece = [&](int slot, int sub, int i) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] + sub * G3_SUB_K),
                                         (lds_void_t*)(smem + slot * G3_SLOT + dst_off + i * 1024), 16, 0, 0);
    };
    auto stage = [&](int slot, int sub) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) stage_piece(slot, sub, i);
    };

    // ---- fragment addresses ----
    const int frow = lane & 15, fgrp = lane >> 4;
    const int pchunk = fgrp ^ (((frow >> 3) & 1) * 2);
    const int a_base = (wr * 128 + frow) * 64 + pchunk * 16;                  // + mi*1024
    const int w_base = G3_PART + (wc * 128 + frow) * 64 + pchunk * 16;        // + ni*1024

    f32x4 acc[2][8][4];                                       // [column half][mi][ni within the half]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsub = K / G3_SUB_K;
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto read_frags = [&](int slot, frag (&af)[8], frag (&wf)[8]) __attribute__((always_inline)) {
        const char* buf = smem + slot * G3_SLOT;
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *(const frag*)(buf + a_base + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = *(const frag*)(buf + w_base + j * 1024);
    };

    // One k-step: the fragments of sub-tile p are in (af, wf); (an, wn) receive those of sub-tile p+1.  With one
    // wave per SIMD nothing else hides a wave's own issue slots, so the 16 fragment reads and the 8 LDS-DMA pieces
    // of the step are spread between its 64 MFMAs (a 16-cycle MFMA leaves room for them) instead of preceding them.
    auto step = [&](int p, int slot, frag (&af)[8], frag (&wf)[8], frag (&an)[8], frag (&wn)[8]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // of this wave's loads only sub-tiles p+2, p+3 may still be in flight
        barrier();                                            // everyone's pieces of p+1 are in; everyone is done reading slot p
        const char* nbuf = smem + (slot + 1 == G4_NSLOT ? 0 : slot + 1) * G3_SLOT;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int idx = 0; idx < 64; ++idx) {
            const int i = idx >> 3, j = idx & 7;
            mfma_agpr<IS_F16>(acc[j >> 2][i][j & 3], wf[j], af[i]);
            if ((idx & 3) == 1) {
                const int r = idx >> 2;
                if (r < 8) an[r] = *(const frag*)(nbuf + a_base + r * 1024);
                else       wn[r - 8] = *(const frag*)(nbuf + w_base + (r - 8) * 1024);
            }
            if ((idx & 7) == 3) stage_piece(slot, p + G4_NSLOT, idx >> 3);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // The last NSLOT (+1) k-steps have nothing left to stage (and the very last nothing to read): plain order.
    auto tail_step = [&](int p, int slot, frag (&af)[8], frag (&wf)[8], frag (&an)[8], frag (&wn)[8]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();
        if (p + 1 < nsub) read_frags(slot + 1 == G4_NSLOT ? 0 : slot + 1, an, wn);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) mfma_agpr<IS_F16>(acc[j >> 2][i][j & 3], wf[j], af[i]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // prologue: sub-tiles 0..NSLOT-1 in flight; sub-tile 0 landed and read
#pragma unroll
    for (int i = 0; i < G4_NSLOT; ++i)
        if (i < nsub) stage(i, i);
    if (nsub >= G4_NSLOT) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else                  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    frag f0a[8], f0w[8], f1a[8], f1w[8];
    read_frags(0, f0a, f0w);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    int slot = 0, p = 0;
    auto next = [&]() { slot = slot + 1 == G4_NSLOT ? 0 : slot + 1; ++p; };
    while (p + G4_NSLOT + 1 < nsub) {                         // steady state, two k-steps per trip (nsub is even)
        step(p, slot, f0a, f0w, f1a, f1w); next();
        step(p, slot, f1a, f1w, f0a, f0w); next();
    }
    while (p < nsub) {
        tail_step(p, slot, f0a, f0w, f1a, f1w); next();
        tail_step(p, slot, f1a, f1w, f0a, f0w); next();
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // the last MFMAs retire before their accumulators are read
    barrier();                                                // all fragment reads retired before LDS becomes the epilogue strips
    char* strip = smem + wave * EPI_WAVE_BYTES;
    wave_epilogue<8>(strip, acc[0], m0 + wr * 128, n0 + wc * 128, lane, epi);
    wave_epilogue<8>(strip, acc[1], m0 + wr * 128, n0 + wc * 128 + 64, lane, epi);
}

// ---- the same tile with the operands staged through registers ------------------------------------------------
// Measured on 16384x4096x4096: the kernel above without its LDS-DMA runs at 1368 TFLOP/s, with it at 872 — with one
// wave per SIMD a global_load_lds occupies the wave's issue for tens of cycles and nothing else can feed the MFMA
// pipe meanwhile.  Here a sub-tile travels global -> VGPR (plain 16-byte loads, issued during k-step s-3) -> LDS
// (ds_write_b128 during k-step s-2) -> fragments (read during s-1) -> MFMAs (s); each of those instructions issues
// in a few cycles between two MFMAs.  Three 32-KiB slots suffice: a slot is written two steps after its last read.
constexpr int G4R_NSLOT = 3;
constexpr int G4R_LDS_BYTES = G4R_NSLOT * G3_SLOT;

template <bool IS_F16, class Epi>
__global__ __launch_bounds__(G4_THREADS, 1)
void gemm_tn256w4r_kernel(const uint16_t* __restrict__ A, int lda,
                          const uint16_t* __restrict__ W, int ldw,
                          int K, int tiles_n, Epi epi) {
    typedef typename mfma_op<IS_F16>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * G2_BM;
    const int n0 = (wg % tiles_n) * G2_BN;

    // thread t moves 16-byte chunks q = t + 256 i (i < 4) of the A part and of the W part of every sub-tile:
    // row = q >> 2 = (t >> 2) + 64 i, logical chunk = t & 3; in LDS the chunk sits at (chunk ^ 2*((row >> 3) & 1)).
    const int lrow = tid >> 2, lchunk = tid & 3;
    const uint16_t* a_src = A + (size_t)(m0 + lrow) * lda + lchunk * 8;
    const uint16_t* w_src = W + (size_t)(n0 + lrow) * ldw + lchunk * 8;
    const size_t a_step = (size_t)64 * lda, w_step = (size_t)64 * ldw;
    const int l_off = lrow * 64 + ((lchunk ^ (((lrow >> 3) & 1) * 2)) * 16);            // + i * 4096 (+ G3_PART for W)

    auto load_chunk = [&](int sub, int c) __attribute__((always_inline)) -> uint4 {      // c: 0-3 A, 4-7 W
        return c < 4 ? *(const uint4*)(a_src + (size_t)c * a_step + sub * G3_SUB_K)
                     : *(const uint4*)(w_src + (size_t)(c - 4) * w_step + sub * G3_SUB_K);
    };
    auto write_chunk = [&](int slot, int c, const uint4& v) __attribute__((always_inline)) {
        *(uint4*)(smem + slot * G3_SLOT + (c < 4 ? 0 : G3_PART) + l_off + (c & 3) * 4096) = v;
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int pchunk = fgrp ^ (((frow >> 3) & 1) * 2);
    const int a_base = (wr * 128 + frow) * 64 + pchunk * 16;
    const int w_base = G3_PART + (wc * 128 + frow) * 64 + pchunk * 16;

    f32x4 acc[2][8][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsub = K / G3_SUB_K;
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto read_frags = [&](int slot, frag (&af)[8], frag (&wf)[8]) __attribute__((always_inline)) {
        const char* buf = smem + slot * G3_SLOT;
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *(const frag*)(buf + a_base + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = *(const frag*)(buf + w_base + j * 1024);
    };
    auto slot_of = [](int sub) { return sub % G4R_NSLOT; };

    // steady-state k-step p (p + 3 < nsub): MFMAs on (af, wf) = sub-tile p; between them
    //   sc (sub-tile p+2, loaded during step p-1) -> LDS;  sub-tile p+3 -> sn;  fragments of p+1 -> (an, wn)
    auto step = [&](int p, frag (&af)[8], frag (&wf)[8], frag (&an)[8], frag (&wn)[8], uint4 (&sc)[8], uint4 (&sn)[8])
        __attribute__((always_inline)) {
        barrier();                                            // sub-tile p+1 is complete in LDS; slot of p+2 is free
        const char* nbuf = smem + slot_of(p + 1) * G3_SLOT;
        const int wslot = slot_of(p + 2);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int idx = 0; idx < 64; ++idx) {
            const int i = idx >> 3, j = idx & 7;
            mfma_agpr<IS_F16>(acc[j >> 2][i][j & 3], wf[j], af[i]);
            if (idx < 16 && (idx & 1) == 0) sn[idx >> 1] = load_chunk(p + 3, idx >> 1);             // 8 global loads: 1.5 steps to land
            if (idx < 32 && (idx & 1) == 1) {                                                       // 16 fragment reads, done well before the step ends
                const int r = idx >> 1;
                if (r < 8) an[r] = *(const frag*)(nbuf + a_base + r * 1024);
                else       wn[r - 8] = *(const frag*)(nbuf + w_base + (r - 8) * 1024);
            }
            if (idx >= 32 && idx < 48 && (idx & 1) == 0)                                            // 8 LDS writes of the sub-tile loaded a step ago
                write_chunk(wslot, (idx - 32) >> 1, sc[(idx - 32) >> 1]);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // last k-steps: whatever of {write p+2, read p+1} still exists, plain order
    auto tail_step = [&](int p, frag (&af)[8], frag (&wf)[8], frag (&an)[8], frag (&wn)[8], uint4 (&sc)[8]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();
        if (p + 2 < nsub) {
#pragma unroll
            for (int c = 0; c < 8; ++c) write_chunk(slot_of(p + 2), c, sc[c]);
        }
        if (p + 1 < nsub) read_frags(slot_of(p + 1), an, wn);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) mfma_agpr<IS_F16>(acc[j >> 2][i][j & 3], wf[j], af[i]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // prologue: sub-tiles 0 and 1 into LDS, sub-tile 2 into s0
    uint4 s0[8], s1[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s0[c] = load_chunk(0, c);
    if (nsub > 1) {
#pragma unroll
        for (int c = 0; c < 8; ++c) s1[c] = load_chunk(1, c);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) write_chunk(0, c, s0[c]);
    if (nsub > 1) {
#pragma unroll
        for (int c = 0; c < 8; ++c) write_chunk(1, c, s1[c]);
    }
    if (nsub > 2) {
#pragma unroll
        for (int c = 0; c < 8; ++c) s0[c] = load_chunk(2, c);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    frag f0a[8], f0w[8], f1a[8], f1w[8];
    read_frags(0, f0a, f0w);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    int p = 0;
    for (; p + 4 < nsub; p += 2) {                            // two k-steps per trip (nsub is even)
        step(p, f0a, f0w, f1a, f1w, s0, s1);
        step(p + 1, f1a, f1w, f0a, f0w, s1, s0);
    }
    for (; p < nsub; p += 2) {
        tail_step(p, f0a, f0w, f1a, f1w, s0);
        if (p + 3 < nsub) {
#pragma unroll
            for (int c = 0; c < 8; ++c) s1[c] = load_chunk(p + 3, c);
        }
        tail_step(p + 1, f1a, f1w, f0a, f0w, s1);
        if (p + 4 < nsub) {
#pragma unroll
            for (int c = 0; c < 8; ++c) s0[c] = load_chunk(p + 4, c);
        }
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    barrier();
    char* strip = smem + wave * EPI_WAVE_BYTES;
    wave_epilogue<8>(strip, acc[0], m0 + wr * 128, n0 + wc * 128, lane, epi);
    wave_epilogue<8>(strip, acc[1], m0 + wr * 128, n0 + wc * 128 + 64, lane, epi);
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256w4(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                               int M, int N, int K, const Epi& epi) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % 64 == 0 && K >= 64,
             "gemm_tn256w4: shape M=%d N=%d K=%d is not tile-aligned (256/256/64)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn256w4: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256w4_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_BYTES));
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256w4r_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G4R_LDS_BYTES));
        attr_set = true;
    }
    static const bool dma = getenv("VQ_AMD_GEMM_W4_DMA") && atoi(getenv("VQ_AMD_GEMM_W4_DMA")) == 1;
    if (dma)
        hipLaunchKernelGGL((gemm_tn256w4_kernel<IS_F16, Epi>), dim3((M / G2_BM) * (N / G2_BN)), dim3(G4_THREADS), G4_LDS_BYTES, st,
                           A, lda, W, ldw, K, N / G2_BN, epi);
    else
        hipLaunchKernelGGL((gemm_tn256w4r_kernel<IS_F16, Epi>), dim3((M / G2_BM) * (N / G2_BN)), dim3(G4_THREADS), G4R_LDS_BYTES, st,
                           A, lda, W, ldw, K, N / G2_BN, epi);
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace vq

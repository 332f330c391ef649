// C[M,N] = A[M,K] * W[N,K]^T with 160x256 output tiles — the ring mainloop of gemm_mfma256.h re-cut for the
// N = 768 GEMMs of the B/32 tower (out_proj, fc2).  At 12,800 rows those have 50 x 3 = 150 tiles of 256x256,
// one wave of workgroups on 150 of the 256 CUs; 160-row tiles give 80 x 3 = 240 workgroups, one per CU on 240
// CUs, each with 0.625 of the work.
//
// Geometry: 8 waves as 2 (rows) x 4 (columns); a wave owns 80 x 64 of the tile = 5 x 4 MFMA 16x16x32 blocks
// (80 accumulator VGPRs).  K advances in 32-element sub-tiles through a 5-slot LDS ring; a slot holds
// 160 A rows + 256 W rows of 64 B (26 KiB, 130 KiB in all).  A sub-tile is 26 LDS-DMA pieces of 16 rows x 64 B;
// wave w issues pieces w, w+8, w+16 (and w+24 for w < 2), so waves 0-1 count 4 loads per sub-tile and the
// others 3 — the vmcnt a wave waits on follows its own count.  Swizzle, stagger between the two row groups and
// the two barriers per phase are those of gemm_tn256_ring_kernel.
#pragma once
#include "gemm_mfma256.h"

namespace vq {

constexpr int G5_BM = 160, G5_BN = 256;
constexpr int G5_AP = G5_BM / 16;                       // 10 A pieces
constexpr int G5_NP = G5_AP + G5_BN / 16;               // 26 pieces per sub-tile
constexpr int G5_SLOT = G5_NP * 1024;                   // 26 KiB
constexpr int G5_NSLOT = 5;
constexpr int G5_LDS_BYTES = G5_NSLOT * G5_SLOT;        // 130 KiB

template <bool IS_F16, class Epi>
__global__ __launch_bounds__(G2_THREADS, 2)
void gemm_tn160_ring_kernel(const uint16_t* __restrict__ A, int lda,
                            const uint16_t* __restrict__ W, int ldw,
                            int K, int tiles_n, Epi epi) {
    typedef mfma_op<IS_F16> op;
    typedef typename op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * G5_BM;
    const int n0 = (wg % tiles_n) * G5_BN;

    // LDS-DMA sources: piece p < 10 is A rows 16p.., piece p >= 10 is W rows 16(p-10)..; its LDS home is p KiB
    // into the slot either way (the A region is followed directly by the W region).
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ (((lane >> 5) & 1) * 2);
    const bool four = wave < G5_NP - 24;                 // waves 0,1 own a fourth piece
    const uint16_t* src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = wave + 8 * i;
        if (p < G5_AP)      src[i] = A + (size_t)(m0 + p * 16 + srow) * lda + schunk * 8;
        else if (p < G5_NP) src[i] = W + (size_t)(n0 + (p - G5_AP) * 16 + srow) * ldw + schunk * 8;
        else                src[i] = W;                  // never issued
    }
    const int piece_off = wave * 1024;

    auto stage = [&](int slot, int sub) __attribute__((always_inline)) {
        char* dst = smem + slot * G5_SLOT + piece_off;
        const int koff = sub * G3_SUB_K;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[0] + koff), (lds_void_t*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[1] + koff), (lds_void_t*)(dst + 8 * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[2] + koff), (lds_void_t*)(dst + 16 * 1024), 16, 0, 0);
        if (four)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[3] + koff), (lds_void_t*)(dst + 24 * 1024), 16, 0, 0);
    };
    // wait until at most `groups` of this wave's sub-tile loads are still in flight
    auto wait_groups3 = [&]() __attribute__((always_inline)) {
        if (four) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int pchunk = fgrp ^ (((frow >> 3) & 1) * 2);
    const int a_base = (wr * (G5_BM / 2) + frow) * 64 + pchunk * 16;               // + mi*1024
    const int w_base = G5_AP * 1024 + (wc * 64 + frow) * 64 + pchunk * 16;         // + ni*1024

    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsub = K / G3_SUB_K;
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    auto phase = [&](int p, int slot, int slot_refill) __attribute__((always_inline)) {
        const char* buf = smem + slot * G5_SLOT;
        frag af[5], wf[4];
#pragma unroll
        for (int i = 0; i < 5; ++i) af[i] = *(const frag*)(buf + a_base + i * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const frag*)(buf + w_base + j * 1024);
        if (p + G5_NSLOT - 1 < nsub) {
            stage(slot_refill, p + G5_NSLOT - 1);
            wait_groups3();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = op::run(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        barrier();
    };

    // prologue: sub-tiles 0..3 in flight, 0 landed
#pragma unroll
    for (int i = 0; i < G5_NSLOT - 1; ++i) stage(i, i);
    wait_groups3();
    barrier();

    if (wr == 1) barrier();               // stagger: row group 1 runs one barrier behind group 0
    int slot = 0, slot_refill = G5_NSLOT - 1;
    for (int p = 0; p < nsub; ++p) {
        phase(p, slot, slot_refill);
        slot_refill = slot;
        slot = slot + 1 == G5_NSLOT ? 0 : slot + 1;
    }
    if (wr == 0) barrier();

    barrier();                            // every wave's fragment reads are retired before LDS becomes the epilogue strip
    wave_epilogue<5>(smem + wave * EPI_WAVE_BYTES, acc, m0 + wr * (G5_BM / 2), n0 + wc * 64, lane, epi);
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn160_ring(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                                  int M, int N, int K, const Epi& epi) {
    VQ_CHECK(M > 0 && M % G5_BM == 0 && N % G5_BN == 0 && K % G3_SUB_K == 0 && K >= (G5_NSLOT - 1) * G3_SUB_K,
             "gemm_tn160_ring: shape M=%d N=%d K=%d is not tile-aligned (160/256/32, K >= 128)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn160_ring: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn160_ring_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn160_ring_kernel<IS_F16, Epi>), dim3((M / G5_BM) * (N / G5_BN)), dim3(G2_THREADS),
                       G5_LDS_BYTES, st, A, lda, W, ldw, K, N / G5_BN, epi);
    VQ_HIP(hipGetLastError());
    return 0;
}

static inline bool gemm_use160() {            // $VQ_AMD_GEMM160=0 keeps the 256x256 kernel for every shape
    static int v = -1;
    if (v < 0) { const char* e = getenv("VQ_AMD_GEMM160"); v = (e && atoi(e) == 0) ? 0 : 1; }
    return v != 0;
}

// True when 160-row tiles fill more CUs than 256-row tiles in a single wave of workgroups.
static inline bool prefer_tn160(int M, int N, int K) {
    if (M % G5_BM || N % G5_BN || K % G3_SUB_K || K < (G5_NSLOT - 1) * G3_SUB_K) return false;
    const int64_t t160 = (int64_t)(M / G5_BM) * (N / G5_BN);
    const int64_t t256 = (int64_t)((M + G2_BM - 1) / G2_BM) * (N / G2_BN);
    return t160 <= 256 && t256 < 200 && t160 > t256;
}

}  // namespace vq

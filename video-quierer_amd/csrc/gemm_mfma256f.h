// 256x256 bf16/fp16 TN GEMM, fourth mainloop: K in 32-wide sub-tiles through a 4-slot LDS ring (the ring kernel's
// LDS image, gemm_mfma256.h), ONE barrier per sub-tile, and the fragments of sub-tile p+1 read into a second
// register set WHILE the 32 MFMAs of sub-tile p run — so the first MFMA after a barrier never waits for LDS.
//
// What the four-phase kernels pay per 16-MFMA phase (s_memtime stamps, r01; PMC r02: SQ_WAIT_ANY 31 % of wave
// time, MFMA pipe 52 % busy inside the K loop): ~145 cycles in which the first MFMA waits for the fragments read
// just before the barrier, and ~150 cycles of barrier skew between the two staggered wave groups — against 256
// cycles of MFMA.  Here a wave's iteration is
//
//     8 MFMAs on register set p&1           (their reads were issued one iteration ago)
//     issue 12 ds_read_b128: fragments of sub-tile p+1 -> register set (p+1)&1
//     issue  4 LDS-DMA pieces: sub-tile p+3 -> ring slot (p+3)&3
//     24 MFMAs on register set p&1
//     s_waitcnt vmcnt(4)                    (sub-tile p+2 has landed; p+3 stays in flight)
//     s_barrier
//
// 8 waves = 2 (m) x 4 (n), 128 x 64 per wave (128 accumulator VGPRs) + 2 x 48 fragment VGPRs.
// Hazards: WAR  slot (p+3)&3 held sub-tile p-1, whose reads were issued in iteration p-2 and had returned before
//               that wave's MFMAs of iteration p-1 (the compiler's counted lgkmcnt), i.e. before the barrier that
//               ends iteration p-1; the refill is issued after that barrier.
//          RAW  sub-tile p+2 is awaited (vmcnt) by every issuing wave before the barrier that ends iteration p and
//               first read in iteration p+1.
// Requirements: M % 256 == 0, N % 256 == 0, K % 128 == 0.
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "gemm_mfma256d.h"

namespace vq {

template <bool IS_F16, class Epi>
__global__ __launch_bounds__(G2_THREADS, 2)
void gemm_tn256f_kernel(const uint16_t* __restrict__ A, int lda,
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
    const int m0 = (wg / tiles_n) * G2_BM;
    const int n0 = (wg % tiles_n) * G2_BN;

    // LDS-DMA: a 1-KiB piece = 16 rows x 64 B; wave w fills pieces 2w, 2w+1 (rows 32w..32w+31) of A and of W
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ (((lane >> 5) & 1) * 2);      // logical chunk stored at physical slot lane&3
    const uint16_t* a_src[2];
    const uint16_t* w_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 16 + srow;
        a_src[i] = A + (size_t)(m0 + row) * lda + schunk * 8;
        w_src[i] = W + (size_t)(n0 + row) * ldw + schunk * 8;
    }
    const int piece_off = wave * 2048;
    auto stage = [&](int slot, int sub) {
        char* dst = smem + slot * G3_SLOT + piece_off;
        const int koff = sub * G3_SUB_K;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[0] + koff), (lds_void_t*)(dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[1] + koff), (lds_void_t*)(dst + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[0] + koff), (lds_void_t*)(dst + G3_PART), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[1] + koff), (lds_void_t*)(dst + G3_PART + 1024), 16, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int pchunk = fgrp ^ (((frow >> 3) & 1) * 2);
    const int a_base = (wr * 128 + frow) * 64 + pchunk * 16;                 // + mi*1024
    const int w_base = G3_PART + (wc * 64 + frow) * 64 + pchunk * 16;        // + ni*1024

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag af0[8], wf0[4], af1[8], wf1[4];

    const int nsub = K / G3_SUB_K;       // even, >= 4
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#define VQ_READ_FRAGS(AF, WF, SLOT)                                                                       \
    do {                                                                                                  \
        const char* b__ = smem + (SLOT) * G3_SLOT;                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) AF[i] = *(const frag*)(b__ + a_base + i * 1024);    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) WF[j] = *(const frag*)(b__ + w_base + j * 1024);    \
    } while (0)
#define VQ_MFMA_ROWS(AF, WF, I0, I1)                                                                      \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = I0; i < I1; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = op::run(WF[j], AF[i], acc[i][j]);   \
    } while (0)

    // prologue: sub-tiles 0, 1, 2 in flight; 0 landed -> its fragments into set 0; 1 landed
    stage(0, 0); stage(1, 1); stage(2, 2);
    const Epi epi_wg = epi_bind_rowstats<G2_BM>(epi, (float2*)(smem + G2_LDS_BYTES), m0, tid, G2_THREADS);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    barrier();
    VQ_READ_FRAGS(af0, wf0, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    barrier();

    // iteration p (two per trip: static register sets): slot of sub-tile q is q & 3.  The prefetch reads and the DMA
    // are issued AFTER the first 8 MFMAs of the cluster: hipcc waits lgkmcnt(0) (not a counted wait) in front of the
    // first MFMA that uses LDS data, which would otherwise also wait for the prefetch it was just handed.
    for (int p = 0; p < nsub; p += 2) {
        const int s0 = p & 3;                  // p is even: s0 in {0, 2}
        // ---- sub-tile p on set 0; prefetch p+1 into set 1 ----
        __builtin_amdgcn_s_setprio(1);
        VQ_MFMA_ROWS(af0, wf0, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        VQ_READ_FRAGS(af1, wf1, s0 + 1);                                   // p+1 < nsub always (nsub even)
        if (p + 3 < nsub) stage((s0 + 3) & 3, p + 3);
        __builtin_amdgcn_sched_barrier(0);
        VQ_MFMA_ROWS(af0, wf0, 2, 8);
        __builtin_amdgcn_s_setprio(0);
        if (p + 3 < nsub)      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // p+2 landed, p+3 in flight
        else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();
        // ---- sub-tile p+1 on set 1; prefetch p+2 into set 0 ----
        __builtin_amdgcn_s_setprio(1);
        VQ_MFMA_ROWS(af1, wf1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (p + 2 < nsub) VQ_READ_FRAGS(af0, wf0, (s0 + 2) & 3);
        if (p + 4 < nsub) stage(s0, p + 4);
        __builtin_amdgcn_sched_barrier(0);
        VQ_MFMA_ROWS(af1, wf1, 2, 8);
        __builtin_amdgcn_s_setprio(0);
        if (p + 4 < nsub)      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // p+3 landed, p+4 in flight
        else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();
    }
#undef VQ_READ_FRAGS
#undef VQ_MFMA_ROWS
    // the last barrier ended the last iteration: every fragment read has returned, LDS is free for the epilogue strips
    wave_epilogue<8>(smem + wave * EPI_WAVE_BYTES, acc, m0 + wr * 128, n0 + wc * 64, lane, epi_wg);
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256f(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                              int M, int N, int K, const Epi& epi) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % 128 == 0,
             "gemm_tn256f: shape M=%d N=%d K=%d is not tile-aligned (256/256/128)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn256f: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256f_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES + G2_ROWSTAT_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn256f_kernel<IS_F16, Epi>), dim3((M / G2_BM) * (N / G2_BN)), dim3(G2_THREADS),
                       G2_LDS_BYTES + (epi_row_in<Epi>::value ? G2_ROWSTAT_BYTES : 0), st, A, lda, W, ldw, K, N / G2_BN, epi);
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace vq

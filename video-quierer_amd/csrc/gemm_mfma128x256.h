// 128(m) x 256(n) bf16/fp16 TN GEMM, TWO workgroups per CU.
//
// Why: s_memtime stamps of the 256x256 kernel inside the tower (scripts/gemm_tower_stamps.py) show a
// workgroup of a K = 768 GEMM spending 29-53 % of its life outside the K loop — ~5k cycles until its first operands
// have landed, 8-34k in the epilogue (a burst of stores, of residual reads, of exp/rcp), against 33k in the loop —
// and a 256x256 workgroup owns its CU (128 KiB of LDS, every register): nothing computes meanwhile.  Here a workgroup is
// 4 waves (one per SIMD) with a 72 KiB ring, so two of them — of the same launch or of another stream's — share a
// CU: one's prologue / epilogue runs under the other's MFMAs, and the two waves of a SIMD belong to different
// workgroups, so they are not in lockstep (the two-waves-same-program collision of kernel 10 does not arise).
//
// Geometry: wave w owns all 128 rows x columns 64w..64w+63 (8 x 4 MFMA 16x16x32 tiles, 128 accumulator VGPRs).  K
// runs in 32-wide sub-tiles through a 3-slot LDS ring (slot = A 128 rows x 64 B | W 256 rows x 64 B = 24 KiB; 64-byte
// rows, chunk c of row r stored at c ^ 2((r>>3)&1): the conflict-free image of gemm_tn256_ring_kernel).  Fragments are
// register double-buffered: while the 32 MFMAs of sub-tile p run, the 12 ds_read_b128 of sub-tile p+1 and the 6
// LDS-DMA pieces (buffer_load ... lds) of sub-tile p+3 are issued; ONE barrier per sub-tile.
//   RAW  sub-tile p+2 is awaited (vmcnt) by every issuing wave before the barrier that ends iteration p; its fragments
//        are read in iteration p+1.
//   WAR  slot (p+3)%3 = p%3 held sub-tile p, whose fragment reads were issued in iteration p-1 and have returned
//        (lgkmcnt(0)) before the barrier that ends iteration p-1; the refill is issued after that barrier.
// W rows are private to the wave that stages them (wave w stages and reads W rows 64w..64w+63); A is shared.
// Requirements: M % 128 == 0, N % 256 == 0, K % 64 == 0, K >= 128, lda/ldw % 8 == 0.
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"

namespace vq {

constexpr int G12_BM = 128, G12_BN = 256, G12_SUB_K = 32, G12_THREADS = 256;
constexpr int G12_A_BYTES = G12_BM * G12_SUB_K * 2;            // 8 KiB
constexpr int G12_SLOT = (G12_BM + G12_BN) * G12_SUB_K * 2;    // 24 KiB
constexpr int G12_NSLOT = 3;
constexpr int G12_LDS_BYTES = G12_NSLOT * G12_SLOT;            // 72 KiB
constexpr int G12_ROWSTAT_BYTES = G12_BM * 8;

template <bool IS_F16, class Epi>
__global__ __launch_bounds__(G12_THREADS, 2)
void gemm_tn128x256_kernel(const uint16_t* __restrict__ A, int lda,
                           const uint16_t* __restrict__ W, int ldw,
                           int K, int tiles_n, Epi epi) {
    typedef mfma_op<IS_F16> op;
    typedef typename op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * G12_BM;
    const int n0 = (wg % tiles_n) * G12_BN;

    // LDS-DMA: a 1-KiB piece = 16 rows x 64 B.  Wave w fills A pieces 2w, 2w+1 (rows 32w..32w+31) and W pieces 4w..4w+3
    // (rows 64w..64w+63: its own columns).  One lane offset per operand; piece row offsets and the K offset are scalar.
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ (((lane >> 5) & 1) * 2);      // logical chunk stored at physical slot lane&3
    const int a_v = ((wave * 32 + srow) * lda + schunk * 8) * 2;
    const int w_v = ((wave * 64 + srow) * ldw + schunk * 8) * 2;
    const int a_p16 = 16 * lda * 2, w_p16 = 16 * ldw * 2;         // 16 source rows, bytes
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, 0x7fffffff, 0x00020000);
    const int a_dst = wave * 2048, w_dst = G12_A_BYTES + wave * 4096;

    auto stage = [&](int slot, int sub) __attribute__((always_inline)) {
        char* base = smem + slot * G12_SLOT;
        const int koff = __builtin_amdgcn_readfirstlane(sub * (G12_SUB_K * 2));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + a_dst), 16, a_v, koff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + a_dst + 1024), 16, a_v, koff + a_p16, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst), 16, w_v, koff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst + 1024), 16, w_v, koff + w_p16, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst + 2048), 16, w_v, koff + 2 * w_p16, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst + 3072), 16, w_v, koff + 3 * w_p16, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int pchunk = fgrp ^ (((frow >> 3) & 1) * 2);
    const int a_base = frow * 64 + pchunk * 16;                              // + mi*1024
    const int w_base = G12_A_BYTES + (wave * 64 + frow) * 64 + pchunk * 16;  // + ni*1024

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag af0[8], wf0[4], af1[8], wf1[4];

    const int nsub = K / G12_SUB_K;       // even, >= 4
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#define VQ_READ_FRAGS(AF, WF, SLOT)                                                                       \
    do {                                                                                                  \
        const char* b__ = smem + (SLOT) * G12_SLOT;                                                       \
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
    const Epi epi_wg = epi_bind_rowstats<G12_BM>(epi, (float2*)(smem + G12_LDS_BYTES), m0, tid, G12_THREADS);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    barrier();
    VQ_READ_FRAGS(af0, wf0, 0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();

    // iteration p (two per trip: static register sets); slot of sub-tile q is q % 3, tracked as a scalar
    int s = 0;                                  // slot of sub-tile p
    for (int p = 0; p < nsub; p += 2) {
        const int s1 = s == 2 ? 0 : s + 1, s2 = s1 == 2 ? 0 : s1 + 1;       // slots of p+1, p+2
        // ---- sub-tile p on set 0; prefetch the fragments of p+1 into set 1; refill slot s (= slot of p+3) ----
        __builtin_amdgcn_s_setprio(1);
        VQ_MFMA_ROWS(af0, wf0, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        VQ_READ_FRAGS(af1, wf1, s1);                                        // p+1 < nsub always (nsub even)
        if (p + 3 < nsub) stage(s, p + 3);
        __builtin_amdgcn_sched_barrier(0);
        VQ_MFMA_ROWS(af0, wf0, 2, 8);
        __builtin_amdgcn_s_setprio(0);
        if (p + 3 < nsub)      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // p+2 landed, p+3 in flight
        else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();
        // ---- sub-tile p+1 on set 1; prefetch p+2 into set 0; refill slot s1 (= slot of p+4) ----
        __builtin_amdgcn_s_setprio(1);
        VQ_MFMA_ROWS(af1, wf1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (p + 2 < nsub) VQ_READ_FRAGS(af0, wf0, s2);
        if (p + 4 < nsub) stage(s1, p + 4);
        __builtin_amdgcn_sched_barrier(0);
        VQ_MFMA_ROWS(af1, wf1, 2, 8);
        __builtin_amdgcn_s_setprio(0);
        if (p + 4 < nsub)      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // p+3 landed, p+4 in flight
        else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();
        s = s2;
    }
#undef VQ_READ_FRAGS
#undef VQ_MFMA_ROWS
    // the last barrier ended the last iteration: every fragment read has returned, LDS is free for the epilogue strips
    wave_epilogue<8>(smem + wave * EPI_WAVE_BYTES, acc, m0, n0 + wave * 64, lane, epi_wg);
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn128x256(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                                 int M, int N, int K, const Epi& epi) {
    VQ_CHECK(M > 0 && M % G12_BM == 0 && N % G12_BN == 0 && K % (2 * G12_SUB_K) == 0 && K >= 4 * G12_SUB_K,
             "gemm_tn128x256: shape M=%d N=%d K=%d is not tile-aligned (128/256/64)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn128x256: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn128x256_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G12_LDS_BYTES + G12_ROWSTAT_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn128x256_kernel<IS_F16, Epi>), dim3((M / G12_BM) * (N / G12_BN)), dim3(G12_THREADS),
                       G12_LDS_BYTES + (epi_row_in<Epi>::value ? G12_ROWSTAT_BYTES : 0), st, A, lda, W, ldw, K, N / G12_BN, epi);
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace vq

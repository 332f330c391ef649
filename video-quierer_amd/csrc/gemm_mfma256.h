// 256x256x64 "phased" bf16/fp16 TN GEMM for gfx950 (second-generation mainloop).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]      (same contract and epilogue functors as gemm_mfma.h)
//
// Why a second kernel: at 128x128 a tile needs 64 B of L2->LDS traffic per MFMA
// cycle per CU, i.e. the whole chip would need ~34 TB/s of L2 bandwidth at full
// MFMA rate; 256x256 halves that and cuts LDS fragment reads per MFMA by 25 %.
//
// Geometry: 512 threads = 8 waves as 2 (m) x 4 (n); a wave owns 128 x 64 of C =
// 8 x 4 MFMA 16x16x32 tiles (128 accumulator VGPRs).  LDS = 2 K-tile buffers x
// {A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255} x 16 KiB = 128 KiB,
// every half-tile a lane-linear LDS-DMA image with the 16-byte chunk XOR-swizzled
// by (row>>1)&7 on the source address and on the fragment read (conflict-free
// ds_read_b128).
//
// Schedule: each K-tile is 4 phases; a phase = {ds_read the register sub-tile it
// needs, issue ONE half-tile of LDS-DMA prefetch, s_barrier, 16 MFMAs (one
// 64 x 32 quadrant x K=64), s_barrier}; the two wave groups (m halves) run
// staggered by one barrier so one group's LDS reads overlap the other's MFMAs.  Quadrant order (0,0) (0,1) (1,1) (1,0)
// re-reads only one operand sub-tile per phase (12+4+8+4 = 28 ds_read_b128 per 64
// MFMAs).  Prefetch for K-tile t+1 is issued during tile t (A half 1, W half 0,
// W half 1 in phases 1-3) and its A half 0 already in phase 4 of tile t-1 into the
// buffer whose A half 0 was last read in phase 3.  The only VMEM wait in the loop
// is a COUNTED s_waitcnt vmcnt(2) in the read half of phase 4 (one half-tile stays
// in flight across the tile boundary); barriers are raw s_barrier, never
// __syncthreads() (which would drain the DMA queue).
//
// Requirements: M % 256 == 0, N % 256 == 0, K % 128 == 0.
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"
#include <cstdlib>

namespace vq {

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64, G2_THREADS = 512;
constexpr int G2_HALF = 128 * G2_BK * 2;          // 16 KiB: 128 rows x 64 k
constexpr int G2_BUF = 4 * G2_HALF;               // 64 KiB per K-tile buffer
constexpr int G2_LDS_BYTES = 2 * G2_BUF;          // 128 KiB

// STAMP = diagnostic build only (vq_debug_gemm_stamps): lane 0 of every wave of workgroup 0 records
// s_memtime at the three points of each phase into `stamps` (never used by the product path).
constexpr int G2_MAX_STAMPS = 768;
template <bool IS_F16, class Epi, bool STAMP = false, bool DIAG = false>
__global__ __launch_bounds__(G2_THREADS, 2)
void gemm_tn256_kernel(const uint16_t* __restrict__ A, int lda,
                       const uint16_t* __restrict__ W, int ldw,
                       int K, int tiles_n, Epi epi, unsigned long long* __restrict__ stamps = nullptr,
                       int diag = 0 /* STAMP builds: bit0 skip DMA in the loop, bit1 skip ds_reads, bit2 skip MFMAs */,
                       int order2d = 0 /* 1: 2-D blocked tile order (4x8 tiles per 32 consecutive workgroups) */) {
    typedef mfma_op<IS_F16> op;
    typedef typename op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int tm = wg / tiles_n, tn = wg % tiles_n;
    if (order2d) tile_coords(wg, (int)gridDim.x / tiles_n, tiles_n, tm, tn);
    const int m0 = tm * G2_BM;
    const int n0 = tn * G2_BN;

    // ---- LDS-DMA source pointers: wave w fills 1-KiB pieces 2w, 2w+1 of a half-tile ----
    const int srow = lane >> 3, sslot = lane & 7;
    const uint16_t* a_src[2];
    const uint16_t* w_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + srow;            // row inside the 128-row half-tile
        const int chunk = sslot ^ ((row >> 1) & 7);
        a_src[i] = A + (size_t)(m0 + row) * lda + chunk * 8;
        w_src[i] = W + (size_t)(n0 + row) * ldw + chunk * 8;
    }
    const size_t a_half = (size_t)128 * lda, w_half = (size_t)128 * ldw;
    const int piece_off = wave * 2048;                         // pieces 2w,2w+1 are contiguous

    // which: 0 = A half 0, 1 = A half 1, 2 = W half 0, 3 = W half 1
    bool in_loop = false;
    auto stage = [&](int buf, int which, int kt) {
        if constexpr (DIAG) { if (in_loop && (diag & 1)) return; }
        char* dst = smem + buf * G2_BUF + which * G2_HALF + piece_off;
        const int koff = kt * G2_BK;
        if (which < 2) {
            const size_t ho = which ? a_half : 0;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[0] + ho + koff), (lds_void_t*)(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[1] + ho + koff), (lds_void_t*)(dst + 1024), 16, 0, 0);
        } else {
            const size_t ho = (which & 1) ? w_half : 0;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[0] + ho + koff), (lds_void_t*)(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[1] + ho + koff), (lds_void_t*)(dst + 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets ----
    const int frow = lane & 15, fgrp = lane >> 4;
    const int fx = (frow >> 1) & 7;
    const int slot[2] = {((0 + fgrp) ^ fx) * 16, ((4 + fgrp) ^ fx) * 16};
    const int a_base = wr * G2_HALF + frow * 128;                                     // + mi*2048
    const int w_base = 2 * G2_HALF + (wc >> 1) * G2_HALF + ((wc & 1) * 64 + frow) * 128;   // + ni*2048

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag af[4][2], wf[2][2];
    if constexpr (DIAG) {                  // diag builds may skip the loads: keep the registers defined
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[i][0] = frag{}; af[i][1] = frag{}; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { wf[j][0] = frag{}; wf[j][1] = frag{}; }
    }

    const int nk = K / G2_BK;

    auto load_a = [&](const char* buf, int hm) {
        if constexpr (DIAG) { if (diag & 2) return; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                af[i][ks] = *(const frag*)(buf + a_base + (hm * 4 + i) * 2048 + slot[ks]);
    };
    auto load_w = [&](const char* buf, int hn) {
        if constexpr (DIAG) { if (diag & 2) return; }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[j][ks] = *(const frag*)(buf + w_base + (hn * 2 + j) * 2048 + slot[ks]);
    };
    auto mfma_quadrant = [&](int hm, int hn) {
        if constexpr (DIAG) { if (diag & 4) return; }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[hm * 4 + i][hn * 2 + j] = op::run(wf[j][ks], af[i][ks], acc[hm * 4 + i][hn * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    int stamp_i = 0;
    auto stamp = [&]() {
        if constexpr (STAMP) {
            if (blockIdx.x == 0 && stamp_i < G2_MAX_STAMPS) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) stamps[wave * G2_MAX_STAMPS + stamp_i] = t;
                ++stamp_i;
            }
        }
    };

    // One K-tile = 4 phases; a phase = read half {ds_reads, one half-tile of DMA} | barrier |
    // MFMA half {16 MFMAs} | barrier.  The two wave groups (wr = 0 / 1, which share every SIMD
    // pairwise) run STAGGERED by one barrier: while one group is in its MFMA half the other is in
    // its read half, so LDS traffic hides under the partner's matrix work.
    //
    // Hazards under the stagger (group 1 is one barrier behind group 0; b(k) = k-th barrier):
    //  RAW  the counted vmcnt that retires tile kt+1's DMA sits in the READ half of phase 4, i.e.
    //       before b(2p) for group 0 and b(2p+1) for group 1; tile kt+1 is first read after b(2p+1)
    //       (group 0) / b(2p+2) (group 1): every issuer's wait precedes a barrier the reader has passed.
    //  WAR  a half-tile is re-staged >= 2 phases after its last ds_read (A half 1, W halves), except
    //       A half 0 in phase 4 right after its last read in phase 3: A half 0 is read by group 0
    //       only, whose phase-3 reads retire (lgkmcnt(0)) before b(2p-1), and nobody issues the
    //       phase-4 DMA before b(2p-1).
    auto tile = [&](int kt, int bufi) {
        const char* buf = smem + bufi * G2_BUF;
        const bool next = kt + 1 < nk, next2 = kt + 2 < nk;
        // phase 1: quadrant (0,0)
        stamp();
        load_a(buf, 0); load_w(buf, 0);
        if (next) stage(bufi ^ 1, 1, kt + 1);
        barrier();
        stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 0);
        stamp();
        barrier();
        // phase 2: quadrant (0,1)
        stamp();
        load_w(buf, 1);
        if (next) stage(bufi ^ 1, 2, kt + 1);
        barrier();
        stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 1);
        stamp();
        barrier();
        // phase 3: quadrant (1,1)
        stamp();
        load_a(buf, 1);
        if (next) stage(bufi ^ 1, 3, kt + 1);
        barrier();
        stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 1);
        stamp();
        barrier();
        // phase 4: quadrant (1,0); this buffer's A half 0 is dead -> start tile kt+2's A half 0,
        // then retire everything older (tile kt+1 complete) with one half-tile left in flight
        stamp();
        load_w(buf, 0);
        if (next2) { stage(bufi, 0, kt + 2); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        else       { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        barrier();
        stamp();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 0);
        stamp();
        barrier();
    };

    // ---- prologue: tile 0 complete, tile 1's A half 0 in flight ----
    stage(0, 0, 0); stage(0, 1, 0); stage(0, 2, 0); stage(0, 3, 0);
    if (nk > 1) { stage(1, 0, 1); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else        { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    barrier();

    in_loop = true;
    if (wr == 1) barrier();               // stagger: group 1 runs one barrier behind group 0
    for (int kt = 0; kt < nk; kt += 2) {
        tile(kt, 0);
        tile(kt + 1, 1);
    }
    if (wr == 0) barrier();               // every wave executes the same number of barriers
    barrier();                            // both groups past their last fragment reads before LDS is reused

    // ---- epilogue: every wave is past its last fragment read (final barrier) -> LDS strips are free ----
    wave_epilogue<8>(smem + wave * EPI_WAVE_BYTES, acc, m0 + wr * 128, n0 + wc * 64, lane, epi);
}

// ---------------------------------------------------------------------------------------------
// Ring variant: same 256x256 tile / wave layout, but the K loop advances in 32-wide sub-tiles
// through a 4-slot LDS ring (4 x {A 256 rows x 64 B, W 256 rows x 64 B} = 128 KiB).
//   phase p (one per sub-tile): read half  = 12 ds_read_b128 (8 A + 4 W fragments of slot p%4),
//                                            4 LDS-DMA pieces refilling slot (p-1)%4 with sub-tile p+3,
//                                            s_waitcnt vmcnt(8) (sub-tile p+1 landed; 8 pieces stay in
//                                            flight), s_waitcnt lgkmcnt(0), s_barrier
//                              MFMA half  = 32 MFMAs (the wave's whole 128x64 tile x K=32), s_barrier
// Twice the MFMAs per barrier pair of the 4-phase-per-K-tile kernel above (measured there with
// s_memtime stamps: ~200 cycles of barrier/restart per half phase against 256 cycles of MFMA), a
// prefetch distance of three sub-tiles, and every DMA wait counted.  The two wave groups run
// staggered by one barrier.  Hazards (b(k) = k-th barrier; group 0 phase p: pre b(2p), close b(2p+1);
// group 1: pre b(2p+1), close b(2p+2)):
//   WAR  slot (p-1)%4 was last read in phase p-1; both groups retire those reads (lgkmcnt(0)) BEFORE
//        their pre-MFMA barrier, i.e. before b(2p-2) / b(2p-1); the earliest refill is issued after b(2p-1).
//   RAW  sub-tile p+1 is retired by every issuer's vmcnt in the read half of phase p (before b(2p) /
//        b(2p+1)); it is first read after b(2p+1) (group 0) / b(2p+2) (group 1).
// 64-byte LDS rows: chunk c of row r is stored at chunk c ^ (2*((r>>3)&1)) — conflict-free for the four
// ds_read_b128 lane groups (brute-forced against the bank model of MI355X_MICROARCH.md §LDS).
constexpr int G3_SUB_K = 32;
constexpr int G3_PART = 256 * G3_SUB_K * 2;       // 16 KiB: 256 rows x 64 B (one operand of one sub-tile)
constexpr int G3_SLOT = 2 * G3_PART;              // 32 KiB
constexpr int G3_LDS_BYTES = 4 * G3_SLOT;         // 128 KiB

template <bool IS_F16, class Epi, bool DIAG = false, int NSLOT = 4>
__global__ __launch_bounds__(G2_THREADS, 2)
void gemm_tn256_ring_kernel(const uint16_t* __restrict__ A, int lda,
                            const uint16_t* __restrict__ W, int ldw,
                            int K, int tiles_n, Epi epi, int diag = 0) {
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
    const int srow = lane >> 2;                                   // row inside the piece
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

    const int nsub = K / G3_SUB_K;
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // NSLOT ring slots (4 = 128 KiB, 5 = 160 KiB): NSLOT-1 sub-tiles in flight while one is consumed.
    // The slot index is wave-uniform run-time state (scalar adds), so the loop needs no unrolling.
    auto phase = [&](int p, int slot, int slot_refill) __attribute__((always_inline)) {
        const char* buf = smem + slot * G3_SLOT;
        frag af[8], wf[4];
        if (DIAG && (diag & 2)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = frag{};
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = frag{};
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = *(const frag*)(buf + a_base + i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(const frag*)(buf + w_base + j * 1024);
        }
        if (p + NSLOT - 1 < nsub) {
            if (!(DIAG && (diag & 1))) stage(slot_refill, p + NSLOT - 1);
            if constexpr (NSLOT == 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else                      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(DIAG && (diag & 8))) barrier();
        if (!(DIAG && (diag & 4))) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = op::run(wf[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
        if (!(DIAG && (diag & 8))) barrier();
    };

    // prologue: sub-tiles 0..NSLOT-2 in flight, 0 landed
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) stage(i, i);
    if constexpr (NSLOT == 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else                      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    barrier();

    if (wr == 1) barrier();               // stagger: group 1 runs one barrier behind group 0
    int slot = 0, slot_refill = NSLOT - 1;
    for (int p = 0; p < nsub; ++p) {
        phase(p, slot, slot_refill);
        slot_refill = slot;               // the slot just consumed is refilled next phase ... (p-1)%NSLOT
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    if (wr == 0) barrier();

    barrier();                            // group 1's last fragment reads are retired before anyone reuses LDS
    wave_epilogue<8>(smem + wave * EPI_WAVE_BYTES, acc, m0 + wr * 128, n0 + wc * 64, lane, epi);
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256_ring(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                                  int M, int N, int K, const Epi& epi) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % 128 == 0,
             "gemm_tn256_ring: shape M=%d N=%d K=%d is not tile-aligned (256/256/128)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn256_ring: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256_ring_kernel<IS_F16, Epi, false, 5>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 5 * G3_SLOT));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tn256_ring_kernel<IS_F16, Epi, false, 5>), dim3((M / G2_BM) * (N / G2_BN)), dim3(G2_THREADS),
                       5 * G3_SLOT, st, A, lda, W, ldw, K, N / G2_BN, epi, 0);
    VQ_HIP(hipGetLastError());
    return 0;
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256_ring_diag(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                                       int M, int N, int K, const Epi& epi, int diag) {
    VQ_CHECK(M % G2_BM == 0 && N % G2_BN == 0 && K % 128 == 0, "gemm_tn256_ring_diag: shape not tile-aligned");
    if (diag & 16) {      // bit4: 5-slot ring
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256_ring_kernel<IS_F16, Epi, true, 5>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 5 * G3_SLOT));
        hipLaunchKernelGGL((gemm_tn256_ring_kernel<IS_F16, Epi, true, 5>), dim3((M / G2_BM) * (N / G2_BN)), dim3(G2_THREADS),
                           5 * G3_SLOT, st, A, lda, W, ldw, K, N / G2_BN, epi, diag);
    } else {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256_ring_kernel<IS_F16, Epi, true, 4>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 4 * G3_SLOT));
        hipLaunchKernelGGL((gemm_tn256_ring_kernel<IS_F16, Epi, true, 4>), dim3((M / G2_BM) * (N / G2_BN)), dim3(G2_THREADS),
                           4 * G3_SLOT, st, A, lda, W, ldw, K, N / G2_BN, epi, diag);
    }
    VQ_HIP(hipGetLastError());
    return 0;
}

static inline int gemm_order2d() {          // $VQ_AMD_TILE2D=1: 2-D blocked tile order (experiment switch)
    static int v = -1;
    if (v < 0) { const char* e = getenv("VQ_AMD_TILE2D"); v = (e && atoi(e) == 1) ? 1 : 0; }
    return v;
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                             int M, int N, int K, const Epi& epi) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0,
             "gemm_tn256: shape M=%d N=%d K=%d is not tile-aligned (256/256/128)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn256: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;       // per instantiation
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
        attr_set = true;
    }
    const int tiles_m = M / G2_BM, tiles_n = N / G2_BN;
    hipLaunchKernelGGL((gemm_tn256_kernel<IS_F16, Epi>), dim3(tiles_m * tiles_n), dim3(G2_THREADS), G2_LDS_BYTES, st,
                       A, lda, W, ldw, K, tiles_n, epi, (unsigned long long*)nullptr, 0, gemm_order2d());
    VQ_HIP(hipGetLastError());
    return 0;
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256_stamped(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                                     int M, int N, int K, const Epi& epi, unsigned long long* d_stamps, int diag = 0) {
    VQ_CHECK(M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0, "gemm_tn256_stamped: shape not tile-aligned");
    if (d_stamps) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256_kernel<IS_F16, Epi, true, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
        hipLaunchKernelGGL((gemm_tn256_kernel<IS_F16, Epi, true, true>), dim3((M / G2_BM) * (N / G2_BN)), dim3(G2_THREADS),
                           G2_LDS_BYTES, st, A, lda, W, ldw, K, N / G2_BN, epi, d_stamps, diag);
    } else {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256_kernel<IS_F16, Epi, false, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
        hipLaunchKernelGGL((gemm_tn256_kernel<IS_F16, Epi, false, true>), dim3((M / G2_BM) * (N / G2_BN)), dim3(G2_THREADS),
                           G2_LDS_BYTES, st, A, lda, W, ldw, K, N / G2_BN, epi, (unsigned long long*)nullptr, diag);
    }
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace vq

// 256x256x64 bf16/fp16 TN GEMM, third-generation mainloop: the four-phase schedule of gemm_mfma256.h with the
// LDS-DMA prefetch running FIVE phases (1.25 K-tiles) ahead of the fragment reads instead of one to three.
//
// Same tile, wave layout (8 waves = 2 m x 4 n, 128 x 64 per wave), LDS image (2 K-tile buffers x 64 KiB, every
// 8-row x 128-B piece lane-linear, 16-byte chunks XOR-swizzled by (row>>1)&7 on the DMA source and on the read) and
// epilogue as gemm_tn256_kernel.  What changes is WHICH rows one staging unit covers and WHEN it is issued:
//
//   A wave reads, per K-tile, the sub-blocks of ITS OWN 128 x 64 tile: rows hm*64..+63 of its 128 A rows in phase 1
//   (hm = 0) and phase 3 (hm = 1), columns hn*32..+31 of its 64 W rows in phase 1 (hn = 0) and phase 2 (hn = 1);
//   the quadrant order is (0,0) (0,1) (1,1) (1,0) and both W sub-blocks stay in registers, so phase 4 reads nothing.
//   A staging unit is therefore the set of LDS rows that ONE phase reads, over all eight waves (16 KiB, two 1-KiB
//   pieces per wave): A0 = {wr*128 + 0..63}, A1 = {wr*128 + 64..127}, W0 = {wc*64 + 0..31}, W1 = {wc*64 + 32..63}.
//   Unit deaths (last ds_read) per tile: A0 and W0 in phase 1, W1 in phase 2, A1 in phase 3 — the same order in
//   which the next tile needs them — so each unit is refilled two or three phases after it died, for the tile
//   after next (same buffer) or the next tile (other buffer):
//
//       phase 1 of tile t   reads A0(t) W0(t)   issues W1(t+1)      waits until W1(t) has landed
//       phase 2             reads W1(t)         issues A1(t+1)      waits until A1(t) has landed
//       phase 3             reads A1(t)         issues A0(t+2)      -
//       phase 4             -                   issues W0(t+2)      waits until A0(t+1), W0(t+1) have landed
//
//   Every unit is issued 5-6 phases before its first read; about five units (80 KiB) are in flight per CU.  Each
//   wait is a counted s_waitcnt vmcnt(8) in steady state (the four units issued after the awaited one stay in
//   flight); the tail of the K loop uses the exact smaller counts.  Barriers are raw s_barrier; the two wave groups
//   run staggered by one barrier as before.
//
// Hazards (b(k) = k-th barrier; phase P of group 0: read half before b(2P), MFMA half before b(2P+1); group 1 one
// barrier later):
//   RAW  a unit is awaited in the read half of phase P (every issuer, both groups: before b(2P) / b(2P+1)) and first
//        read in the read half of phase P+1 (after b(2P+1) / b(2P+2)): every issuer's wait precedes a barrier the
//        reader has passed.
//   WAR  a unit is re-issued >= 2 phases after its last ds_read; those reads retire (lgkmcnt(0)) in the MFMA half of
//        their phase, i.e. before b(2P+1) / b(2P+2), and the earliest re-issue (phase P+2, group 0) comes after b(2P+3).
//
// Requirements: M % 256 == 0, N % 256 == 0, K % 128 == 0 (two K-tiles per loop trip).
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"

// Cache policy of the LDS-DMA operand loads (the `aux` immediate of buffer_load ... lds: 0 = default, 2 = nt "streamed, first
// to leave L2").  In the multi-tile kernel a workgroup walks `tpw` column tiles of one tile row: its A panel is read again for
// every column tile while each W panel passes through an XCD's L2 once — W marked nt keeps the A panels resident
// (DESIGN.md section 4 "Round 4").  Build-time switches so that variants can be A/B-ed as separate libraries ($VQ_AMD_LIB).
#ifndef VQ_GEMM_DM_W_AUX
#define VQ_GEMM_DM_W_AUX 0
#endif
#ifndef VQ_GEMM_DM_A_AUX
#define VQ_GEMM_DM_A_AUX 0
#endif
#ifndef VQ_GEMM_D_W_AUX
#define VQ_GEMM_D_W_AUX 0
#endif
#ifndef VQ_GEMM_D_A_AUX
#define VQ_GEMM_D_A_AUX 0
#endif

namespace vq {

#ifdef VQ_GEMM_TOWER_STAMPS      // `make STAMPS=1`: diagnostic build for scripts/gemm_tower_stamps.py, never the product library
__device__ unsigned long long g_dbg_stamps[4096 * 8];
__device__ unsigned int g_dbg_count;
#define VQ_TOWER_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define VQ_TOWER_REALTIME(v) const unsigned long long v = __builtin_amdgcn_s_memrealtime()     /* constant 100 MHz */
#else
#define VQ_TOWER_STAMP(v)
#define VQ_TOWER_REALTIME(v)
#endif

constexpr int G2D_STAMPS = 512;                    // per wave, CLOCK == 2 diagnostic builds
constexpr int G2_ROWSTAT_BYTES = G2_BM * 8;      // (mean, rstd) per tile row behind the two K-tile buffers (kRowIn epilogues)

template <bool IS_F16, class Epi, int CLOCK = 0 /* diagnostic builds: 1 = clock around the K loop, 2 = s_memtime stamps per phase */,
          bool BUF = false /* LDS-DMA as buffer_load ... lds (descriptor + 32-bit lane offset + scalar K offset) instead of global_load_lds */,
          bool NARROW = false /* diagnostic TIMING ablation (wrong results): a 256 x 192 tile - every wave's second W sub-block is one MFMA
                                 column tile instead of two (48 MFMAs, 20 fragment reads and 14 LDS-DMA pieces per K-tile and wave instead
                                 of 64 / 24 / 16): what a q|k|v-of-one-head tile would cost in this mainloop */>
__global__ __launch_bounds__(G2_THREADS, 2)
void gemm_tn256d_kernel(const uint16_t* __restrict__ A, int lda,
                        const uint16_t* __restrict__ W, int ldw,
                        int K, int tiles_n, Epi epi, int order2d,
                        unsigned long long* __restrict__ clock_out = nullptr /* CLOCK builds: per-workgroup {d memtime, d memrealtime} around the K loop */) {
    VQ_TOWER_STAMP(ts0);
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

    // ---- LDS-DMA: wave w fills pieces p = 2w, 2w+1 (8 rows x 128 B each) of every staging unit ----
    // A unit hm: piece p -> rows (p>>3)*128 + hm*64 + (p&7)*8 ..+7        (LDS: A region, row*128 B)
    // W unit hn: piece p -> rows (p>>2)*64  + hn*32 + (p&3)*8 ..+7        (LDS: W region, row*128 B)
    const int srow = lane >> 3, sslot = lane & 7;
    const uint16_t* a_src[2][2];
    const uint16_t* w_src[2][2];
    int a_dst[2][2], w_dst[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wave * 2 + i;
            const int arow0 = (p >> 3) * 128 + h * 64 + (p & 7) * 8;
            const int wrow0 = (p >> 2) * 64 + h * 32 + (p & 3) * 8;
            const int ar = arow0 + srow, wrw = wrow0 + srow;
            a_src[h][i] = A + (size_t)(m0 + ar) * lda + (sslot ^ ((ar >> 1) & 7)) * 8;
            w_src[h][i] = W + (size_t)(n0 + wrw) * ldw + (sslot ^ ((wrw >> 1) & 7)) * 8;
            a_dst[h][i] = arow0 * 128;                       // A region = first 32 KiB of a buffer (two 16-KiB halves)
            w_dst[h][i] = 2 * G2_HALF + wrow0 * 128;         // W region = second 32 KiB
        }
    // BUF: one descriptor per operand (base = the tile's first row: workgroup-uniform), the lane's byte offset inside
    // the tile in a VGPR, the K offset in an SGPR — no 64-bit address arithmetic per piece
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, 0x7fffffff, 0x00020000);
    int a_voff[2][2], w_voff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a_voff[h][i] = (int)((a_src[h][i] - (A + (size_t)m0 * lda)) * 2);
            w_voff[h][i] = (int)((w_src[h][i] - (W + (size_t)n0 * ldw)) * 2);
        }

    auto stage_a = [&](int buf, int hm, int kt) {
        char* base = smem + buf * G2_BUF;
        const int koff = kt * G2_BK;
        if constexpr (BUF) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + a_dst[hm][0]), 16, a_voff[hm][0], koff * 2, 0, VQ_GEMM_D_A_AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + a_dst[hm][1]), 16, a_voff[hm][1], koff * 2, 0, VQ_GEMM_D_A_AUX);
        } else {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[hm][0] + koff), (lds_void_t*)(base + a_dst[hm][0]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[hm][1] + koff), (lds_void_t*)(base + a_dst[hm][1]), 16, 0, 0);
        }
    };
    auto stage_w = [&](int buf, int hn, int kt) {
        char* base = smem + buf * G2_BUF;
        const int koff = kt * G2_BK;
        if constexpr (BUF) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst[hn][0]), 16, w_voff[hn][0], koff * 2, 0, VQ_GEMM_D_W_AUX);
            if (!(NARROW && hn == 1))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst[hn][1]), 16, w_voff[hn][1], koff * 2, 0, VQ_GEMM_D_W_AUX);
        } else {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[hn][0] + koff), (lds_void_t*)(base + w_dst[hn][0]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[hn][1] + koff), (lds_void_t*)(base + w_dst[hn][1]), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (identical to gemm_tn256_kernel) ----
    const int frow = lane & 15, fgrp = lane >> 4;
    const int fx = (frow >> 1) & 7;
    const int slot[2] = {((0 + fgrp) ^ fx) * 16, ((4 + fgrp) ^ fx) * 16};
    const int a_base = wr * G2_HALF + frow * 128;                                          // + mi*2048
    const int w_base = 2 * G2_HALF + (wc >> 1) * G2_HALF + ((wc & 1) * 64 + frow) * 128;   // + ni*2048

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag af[4][2], wf[2][2][2];          // one A sub-block (64 rows); BOTH W sub-blocks (2 x 32 cols)

    const int nk = K / G2_BK;

    auto load_a = [&](const char* buf, int hm) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                af[i][ks] = *(const frag*)(buf + a_base + (hm * 4 + i) * 2048 + slot[ks]);
    };
    auto load_w = [&](const char* buf, int hn) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                if (!(NARROW && hn == 1 && j == 1))
                wf[hn][j][ks] = *(const frag*)(buf + w_base + (hn * 2 + j) * 2048 + slot[ks]);
    };
    auto mfma_quadrant = [&](int hm, int hn) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (!(NARROW && hn == 1 && j == 1))
                    acc[hm * 4 + i][hn * 2 + j] = op::run(wf[hn][j][ks], af[i][ks], acc[hm * 4 + i][hn * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#define VQ_VMCNT_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// (n = pieces issued after the awaited unit; the NARROW ablation's W1 unit is one piece per wave, so every count that includes a W1 is one less)
#define VQ_VMCNT(n) do { if constexpr (NARROW) { if (n == 8) VQ_VMCNT_(7); else if (n == 4) VQ_VMCNT_(3); else VQ_VMCNT_(n); } else VQ_VMCNT_(n); } while (0)
    // CLOCK == 2: four s_memtime stamps per phase (start, before the mid barrier, before the MFMAs, after them) kept in
    // the LDS behind the two K-tile buffers (the launch asks for 160 KiB then)
    unsigned long long* stamp_lds = (unsigned long long*)(smem + G2_LDS_BYTES);
    int stamp_i = 0;
    auto stamp = [&]() {
        if constexpr (CLOCK == 2) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (stamp_i < G2D_STAMPS) { if (lane == 0) stamp_lds[wave * G2D_STAMPS + stamp_i] = t; ++stamp_i; }
        }
    };

    auto tile = [&](int kt, int bufi) {
        const char* buf = smem + bufi * G2_BUF;
        const bool next = kt + 1 < nk, next2 = kt + 2 < nk;
        // phase 1: quadrant (0,0)
        stamp();
        load_a(buf, 0); load_w(buf, 0);
        if (next) { stage_w(bufi ^ 1, 1, kt + 1); VQ_VMCNT(8); }       // newer than W1(t): A1(t) A0(t+1) W0(t+1) W1(t+1)
        else      { VQ_VMCNT(2); }                                     //                   A1(t)
        stamp();
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();
        mfma_quadrant(0, 0);
        stamp();
        barrier();
        // phase 2: quadrant (0,1)
        stamp();
        load_w(buf, 1);
        if (next) { stage_a(bufi ^ 1, 1, kt + 1); VQ_VMCNT(8); }       // newer than A1(t): A0(t+1) W0(t+1) W1(t+1) A1(t+1)
        else      { VQ_VMCNT(0); }
        stamp();
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();
        mfma_quadrant(0, 1);
        stamp();
        barrier();
        // phase 3: quadrant (1,1)
        stamp();
        load_a(buf, 1);
        if (next2) stage_a(bufi, 0, kt + 2);
        stamp();
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();
        mfma_quadrant(1, 1);
        stamp();
        barrier();
        // phase 4: quadrant (1,0): no fragment reads (A1 and W0 are in registers)
        stamp();
        if (next2)     { stage_w(bufi, 0, kt + 2); VQ_VMCNT(8); }      // newer than W0(t+1): W1(t+1) A1(t+1) A0(t+2) W0(t+2)
        else if (next) { VQ_VMCNT(4); }                                //                     W1(t+1) A1(t+1)
        stamp();
        barrier();
        stamp();
        mfma_quadrant(1, 0);
        stamp();
        barrier();
    };

    // ---- prologue: tile 0 complete + A0, W0 of tile 1 in flight; A0(0), W0(0) landed ----
    stage_a(0, 0, 0); stage_w(0, 0, 0); stage_w(0, 1, 0); stage_a(0, 1, 0);
    if (nk > 1) { stage_a(1, 0, 1); stage_w(1, 0, 1); }
    // (mean, rstd) of the tile's rows for LayerNorm-consuming epilogues, while the first units are in flight
    const Epi epi_wg = epi_bind_rowstats<G2_BM>(epi, (float2*)(smem + G2_LDS_BYTES), m0, tid, G2_THREADS);
    if (nk > 1) { VQ_VMCNT(8); }
    else        { VQ_VMCNT(4); }
    barrier();

    unsigned long long c0 = 0, r0 = 0;
    if constexpr (CLOCK == 1) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    VQ_TOWER_STAMP(ts1);
    VQ_TOWER_REALTIME(tr1);
    if (wr == 1) barrier();               // stagger: group 1 runs one barrier behind group 0
    for (int kt = 0; kt < nk; kt += 2) {
        tile(kt, 0);
        tile(kt + 1, 1);
    }
    if (wr == 0) barrier();               // every wave executes the same number of barriers
    barrier();                            // both groups past their last fragment reads before LDS is reused
#undef VQ_VMCNT
#undef VQ_VMCNT_
    VQ_TOWER_STAMP(ts2);
    VQ_TOWER_REALTIME(tr2);

    if constexpr (CLOCK == 1) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) { clock_out[blockIdx.x * 2] = c1 - c0; clock_out[blockIdx.x * 2 + 1] = r1 - r0; }
    }
    if constexpr (CLOCK == 2) {           // stamps of workgroup 0: LDS -> clock_out[wave][G2D_STAMPS]
        if (blockIdx.x == 0)
            for (int i = lane; i < G2D_STAMPS; i += 64) clock_out[wave * G2D_STAMPS + i] = stamp_lds[wave * G2D_STAMPS + i];
        __syncthreads();
    }

    wave_epilogue<8>(smem + wave * EPI_WAVE_BYTES, acc, m0 + wr * 128, n0 + wc * 64, lane, epi_wg);
#ifdef VQ_GEMM_TOWER_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VQ_TOWER_STAMP(ts3);
    if (CLOCK == 0 && lane == 0 && wave == 5 && (blockIdx.x % 37) == 5) {      // a sample of workgroups: {K, tiles_n, epilogue tag, cycles x3, K-loop wall ticks, id}
        const unsigned int sl = atomicAdd(&g_dbg_count, 1u) % 4096u;       // a ring: the dump holds the LAST 4096 samples
        unsigned long long* d = g_dbg_stamps + sl * 8;
        d[0] = K; d[1] = tiles_n; d[2] = sizeof(Epi) * 4 + (epi_row_in<Epi>::value ? 1 : 0) + (Epi::kLoads ? 2 : 0);
        d[3] = ts1 - ts0; d[4] = ts2 - ts1; d[5] = ts3 - ts2; d[6] = tr2 - tr1 /* 10 ns ticks over the K loop */; d[7] = blockIdx.x;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-tile form of gemm_tn256d_kernel: a workgroup computes `tpw` consecutive tiles of ONE tile row (tiles_n % tpw == 0)
// and the first K-tile of tile i+1 lands WHILE tile i's epilogue runs.  Stamps: a K = 768 workgroup spends
// ~5k of its ~50k cycles waiting for its first operands, and a 256x256 workgroup owns its CU, so nothing computes meanwhile;
// here only the first tile of a workgroup pays that.  LDS: buffer 0 = [0, 64K) takes the next tile's K-tile 0 during the
// epilogue, whose per-wave transposition strips therefore live in [64K, 64K + 8 x 8704) (buffer 1 and the 5.5 KiB behind
// it), the LayerNorm row statistics behind them (the rows are those of the whole tile row: computed once).
//   after the K loop      every fragment read has retired (final barrier)      -> stage K-tile 0 of the next tile into buffer 0
//   epilogue              strips in buffer 1 (its last reads were the last K-tile's)
//   barrier               every wave is done with its strip                    -> stage A0, W0 of K-tile 1 into buffer 1
//   s_waitcnt vmcnt(4)    K-tile 0 (and the epilogue's stores) retired, the two new units stay in flight; barrier; K loop
// Same mainloop, schedule and hazards as gemm_tn256d_kernel.  lda / ldw % 64 == 0 (four lane-offset registers).
constexpr int G2M_STRIP_OFF = G2_BUF;                                        // strips start at buffer 1
constexpr int G2M_ROWSTAT_OFF = G2_BUF + 8 * EPI_WAVE_BYTES;                 // 135,168
constexpr int G2M_LDS_BYTES = G2M_ROWSTAT_OFF + G2_ROWSTAT_BYTES;            // 137,216

template <bool IS_F16, class Epi>
__global__ __launch_bounds__(G2_THREADS, 2)
void gemm_tn256dm_kernel(const uint16_t* __restrict__ A, int lda,
                         const uint16_t* __restrict__ W, int ldw,
                         int K, int tiles_n, Epi epi, int tpw) {
    typedef mfma_op<IS_F16> op;
    typedef typename op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int groups_n = tiles_n / tpw;                  // workgroups per tile row
    const int tm = wg / groups_n, tn0 = (wg - tm * groups_n) * tpw;
    const int m0 = tm * G2_BM;

    // staging: piece 2w of unit 0 per operand (+ the same ^ 64 for piece 2w+1); unit / piece rows and K offset are scalar
    const int srow = lane >> 3, sslot = lane & 7;
    const int arow_w = (wave >> 2) * 128 + (wave & 3) * 16, wrow_w = (wave >> 1) * 64 + (wave & 1) * 16;
    const int ar = arow_w + srow, wrw = wrow_w + srow;
    const int a_v0 = (ar * lda + (sslot ^ ((ar >> 1) & 7)) * 8) * 2, a_v1 = a_v0 ^ 64;
    const int w_v0 = (wrw * ldw + (sslot ^ ((wrw >> 1) & 7)) * 8) * 2, w_v1 = w_v0 ^ 64;
    const int a_dst0 = arow_w * 128, w_dst0 = 2 * G2_HALF + wrow_w * 128;
    const int a_row8 = 8 * lda * 2, w_row8 = 8 * ldw * 2;
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    int w_tile = 0;                                      // byte offset of the current tile's first W row (scalar)

    auto stage_a = [&](int buf, int hm, int kt) __attribute__((always_inline)) {
        char* base = smem + buf * G2_BUF + a_dst0 + hm * (64 * 128);
        const int soff = __builtin_amdgcn_readfirstlane(kt * (G2_BK * 2) + hm * 8 * a_row8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base), 16, a_v0, soff, 0, VQ_GEMM_DM_A_AUX);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + 1024), 16, a_v1, soff + a_row8, 0, VQ_GEMM_DM_A_AUX);
    };
    auto stage_w_at = [&](int buf, int hn, int kt, int wt) __attribute__((always_inline)) {
        char* base = smem + buf * G2_BUF + w_dst0 + hn * (32 * 128);
        const int soff = __builtin_amdgcn_readfirstlane(wt + kt * (G2_BK * 2) + hn * 4 * w_row8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base), 16, w_v0, soff, 0, VQ_GEMM_DM_W_AUX);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + 1024), 16, w_v1, soff + w_row8, 0, VQ_GEMM_DM_W_AUX);
    };
    auto stage_w = [&](int buf, int hn, int kt) __attribute__((always_inline)) { stage_w_at(buf, hn, kt, w_tile); };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int fx = (frow >> 1) & 7;
    const int slot[2] = {((0 + fgrp) ^ fx) * 16, ((4 + fgrp) ^ fx) * 16};
    const int a_base = wr * G2_HALF + frow * 128;
    const int w_base = 2 * G2_HALF + (wc >> 1) * G2_HALF + ((wc & 1) * 64 + frow) * 128;

    f32x4 acc[8][4];
    frag af[4][2], wf[2][2][2];
    const int nk = K / G2_BK;

    auto load_a = [&](const char* buf, int hm) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                af[i][ks] = *(const frag*)(buf + a_base + (hm * 4 + i) * 2048 + slot[ks]);
    };
    auto load_w = [&](const char* buf, int hn) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[hn][j][ks] = *(const frag*)(buf + w_base + (hn * 2 + j) * 2048 + slot[ks]);
    };
    auto mfma_quadrant = [&](int hm, int hn) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[hm * 4 + i][hn * 2 + j] = op::run(wf[hn][j][ks], af[i][ks], acc[hm * 4 + i][hn * 2 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#define VQ_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

    auto tile = [&](int kt, int bufi) __attribute__((always_inline)) {       // one K-tile: as in gemm_tn256d_kernel
        const char* buf = smem + bufi * G2_BUF;
        const bool next = kt + 1 < nk, next2 = kt + 2 < nk;
        load_a(buf, 0); load_w(buf, 0);
        if (next) { stage_w(bufi ^ 1, 1, kt + 1); VQ_VMCNT(8); }
        else      { VQ_VMCNT(2); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 0);
        barrier();
        load_w(buf, 1);
        if (next) { stage_a(bufi ^ 1, 1, kt + 1); VQ_VMCNT(8); }
        else      { VQ_VMCNT(0); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 1);
        barrier();
        load_a(buf, 1);
        if (next2) stage_a(bufi, 0, kt + 2);
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 1);
        barrier();
        if (next2)     { stage_w(bufi, 0, kt + 2); VQ_VMCNT(8); }
        else if (next) { VQ_VMCNT(4); }
        barrier();
        mfma_quadrant(1, 0);
        barrier();
    };

    // first tile: the whole of K-tile 0 and A0, W0 of K-tile 1; row statistics of the tile row while they are in flight
    w_tile = __builtin_amdgcn_readfirstlane(tn0 * G2_BN * ldw * 2);
    stage_a(0, 0, 0); stage_w(0, 0, 0); stage_w(0, 1, 0); stage_a(0, 1, 0);
    if (nk > 1) { stage_a(1, 0, 1); stage_w(1, 0, 1); }
    const Epi epi_wg = epi_bind_rowstats<G2_BM>(epi, (float2*)(smem + G2M_ROWSTAT_OFF), m0, tid, G2_THREADS);
    if (nk > 1) { VQ_VMCNT(8); }
    else        { VQ_VMCNT(4); }
    barrier();

    for (int it = 0; it < tpw; ++it) {
        const int n0 = (tn0 + it) * G2_BN;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (wr == 1) barrier();               // stagger: group 1 runs one barrier behind group 0
        for (int kt = 0; kt < nk; kt += 2) {
            tile(kt, 0);
            tile(kt + 1, 1);
        }
        if (wr == 0) barrier();               // every wave executes the same number of barriers
        barrier();                            // both groups past their last fragment reads

        const bool more = it + 1 < tpw;
        const int w_next = __builtin_amdgcn_readfirstlane((tn0 + it + 1) * G2_BN * ldw * 2);
        // the next tile's K-tile 0 -> buffer 0, under this tile's epilogue, behind the epilogue's first loads
        auto stage_next = [&]() __attribute__((always_inline)) {
            if (more) { stage_a(0, 0, 0); stage_w_at(0, 0, 0, w_next); stage_w_at(0, 1, 0, w_next); stage_a(0, 1, 0); }
        };
        // the row part of the epilogue's addresses is the same for every tile of the workgroup: left alone the compiler hoists
        // ~20 registers of it across the K loop (and spills them); an opaque move makes it recompute them per tile
        int m_wave = m0 + wr * 128, lane_e = lane;
        asm volatile("" : "+v"(m_wave), "+v"(lane_e));
        wave_epilogue<8>(smem + G2M_STRIP_OFF + wave * EPI_WAVE_BYTES, acc, m_wave, n0 + wc * 64, lane_e, epi_wg, stage_next);
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            barrier();                         // every wave is done with its strip: buffer 1 may be refilled
            w_tile = w_next;
            if (nk > 1) { stage_a(1, 0, 1); stage_w(1, 0, 1); VQ_VMCNT(4); }
            else        { VQ_VMCNT(0); }
            barrier();
        }
    }
#undef VQ_VMCNT
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256dm(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                               int M, int N, int K, const Epi& epi, int tpw) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0 && tpw >= 1 && (N / G2_BN) % tpw == 0,
             "gemm_tn256dm: shape M=%d N=%d K=%d / %d tiles per workgroup is not tile-aligned (256/256/128)", M, N, K, tpw);
    VQ_CHECK(lda % 64 == 0 && ldw % 64 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && (int64_t)N * ldw * 2 < ((int64_t)1 << 31),
             "gemm_tn256dm: operands must be 16-byte aligned with lda/ldw %% 64 == 0 and W below 2 GiB");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256dm_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G2M_LDS_BYTES));
        attr_set = true;
    }
    const int tiles_m = M / G2_BM, tiles_n = N / G2_BN;
    hipLaunchKernelGGL((gemm_tn256dm_kernel<IS_F16, Epi>), dim3(tiles_m * (tiles_n / tpw)), dim3(G2_THREADS), G2M_LDS_BYTES, st,
                       A, lda, W, ldw, K, tiles_n, epi, tpw);
    VQ_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Two-phase variant: the same staging units and LDS image, but a K-tile is TWO phases of 32 MFMAs instead of four
// of 16 — half the barriers (4 per K-tile) and half the points where the first MFMA waits for its fragments:
//
//     phase A of tile t   reads A0 W0 W1 (16 ds_read_b128)   issues A0 W0 W1 of tile t+1   waits until A1(t) has landed
//                         32 MFMAs: quadrants (0,0) (0,1)
//     phase B             reads A1 (8 ds_read_b128)          issues A1 of tile t+1         waits until A0 W0 W1 (t+1) have landed
//                         32 MFMAs: quadrants (1,1) (1,0)
//
// Every unit is re-issued two phases after its last read (WAR) and awaited one phase before its first read (RAW),
// exactly as in the four-phase schedule above; the waits are vmcnt(6) / vmcnt(2) in steady state.
//
// Staging is `buffer_load ... lds` with FOUR lane-offset registers (piece 2w of unit 0 of each operand and the same
// ^ 64 for piece 2w+1, whose swizzle differs by chunk ^ 4; unit / piece row offsets and the K offset are scalar).  The
// first build of this schedule kept eight 64-bit source pointers per lane, three of which the register allocator
// spilled and RELOADED INSIDE THE K LOOP: a scratch load waits vmcnt(0), i.e. for every LDS-DMA unit in flight, so
// what was measured (and rejected) in round 2 was the spill, not the schedule.  Requires lda, ldw % 64 == 0.
template <bool IS_F16, class Epi>
__global__ __launch_bounds__(G2_THREADS, 2)
void gemm_tn256e_kernel(const uint16_t* __restrict__ A, int lda,
                        const uint16_t* __restrict__ W, int ldw,
                        int K, int tiles_n, Epi epi, int order2d) {
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

    // A unit hm: piece p -> rows (p>>3)*128 + hm*64 + (p&7)*8 ..+7;  W unit hn: piece p -> rows (p>>2)*64 + hn*32 + (p&3)*8 ..+7
    const int srow = lane >> 3, sslot = lane & 7;
    const int arow_w = (wave >> 2) * 128 + (wave & 3) * 16, wrow_w = (wave >> 1) * 64 + (wave & 1) * 16;   // piece 2w of unit 0
    const int ar = arow_w + srow, wrw = wrow_w + srow;
    const int a_v0 = (ar * lda + (sslot ^ ((ar >> 1) & 7)) * 8) * 2, a_v1 = a_v0 ^ 64;
    const int w_v0 = (wrw * ldw + (sslot ^ ((wrw >> 1) & 7)) * 8) * 2, w_v1 = w_v0 ^ 64;
    const int a_dst0 = arow_w * 128, w_dst0 = 2 * G2_HALF + wrow_w * 128;
    const int a_row8 = 8 * lda * 2, w_row8 = 8 * ldw * 2;                  // 8 source rows, bytes
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, 0x7fffffff, 0x00020000);

    auto stage_a = [&](int buf, int hm, int kt) __attribute__((always_inline)) {
        char* base = smem + buf * G2_BUF + a_dst0 + hm * (64 * 128);
        const int soff = __builtin_amdgcn_readfirstlane(kt * (G2_BK * 2) + hm * 8 * a_row8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base), 16, a_v0, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + 1024), 16, a_v1, soff + a_row8, 0, 0);
    };
    auto stage_w = [&](int buf, int hn, int kt) __attribute__((always_inline)) {
        char* base = smem + buf * G2_BUF + w_dst0 + hn * (32 * 128);
        const int soff = __builtin_amdgcn_readfirstlane(kt * (G2_BK * 2) + hn * 4 * w_row8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base), 16, w_v0, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + 1024), 16, w_v1, soff + w_row8, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int fx = (frow >> 1) & 7;
    const int slot[2] = {((0 + fgrp) ^ fx) * 16, ((4 + fgrp) ^ fx) * 16};
    const int a_base = wr * G2_HALF + frow * 128;
    const int w_base = 2 * G2_HALF + (wc >> 1) * G2_HALF + ((wc & 1) * 64 + frow) * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag af[4][2], wf[4][2];             // one A sub-block (64 rows); the wave's whole W block (64 cols)

    const int nk = K / G2_BK;
    auto load_a = [&](const char* buf, int hm) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                af[i][ks] = *(const frag*)(buf + a_base + (hm * 4 + i) * 2048 + slot[ks]);
    };
    auto load_w = [&](const char* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[j][ks] = *(const frag*)(buf + w_base + j * 2048 + slot[ks]);
    };
    auto mfma_half = [&](int hm) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[hm * 4 + i][j] = op::run(wf[j][ks], af[i][ks], acc[hm * 4 + i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#define VQ_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

    auto tile = [&](int kt, int bufi) __attribute__((always_inline)) {
        const char* buf = smem + bufi * G2_BUF;
        const bool next = kt + 1 < nk;
        // phase A: rows 0..63 of the wave's tile x all 64 columns
        load_a(buf, 0); load_w(buf);
        if (next) { stage_a(bufi ^ 1, 0, kt + 1); stage_w(bufi ^ 1, 0, kt + 1); stage_w(bufi ^ 1, 1, kt + 1); VQ_VMCNT(6); }
        else      { VQ_VMCNT(0); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_half(0);
        barrier();
        // phase B: rows 64..127
        load_a(buf, 1);
        if (next) { stage_a(bufi ^ 1, 1, kt + 1); VQ_VMCNT(2); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_half(1);
        barrier();
    };

    stage_a(0, 0, 0); stage_w(0, 0, 0); stage_w(0, 1, 0); stage_a(0, 1, 0);
    // (mean, rstd) of the tile's rows for LayerNorm-consuming epilogues, while the first units are in flight
    const Epi epi_wg = epi_bind_rowstats<G2_BM>(epi, (float2*)(smem + G2_LDS_BYTES), m0, tid, G2_THREADS);
    VQ_VMCNT(2);
    barrier();

    if (wr == 1) barrier();               // stagger: group 1 runs one barrier behind group 0
    for (int kt = 0; kt < nk; kt += 2) {
        tile(kt, 0);
        tile(kt + 1, 1);
    }
    if (wr == 0) barrier();
    barrier();
#undef VQ_VMCNT

    wave_epilogue<8>(smem + wave * EPI_WAVE_BYTES, acc, m0 + wr * 128, n0 + wc * 64, lane, epi_wg);
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256e(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                              int M, int N, int K, const Epi& epi) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0,
             "gemm_tn256e: shape M=%d N=%d K=%d is not tile-aligned (256/256/128)", M, N, K);
    VQ_CHECK(lda % 64 == 0 && ldw % 64 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn256e: operands must be 16-byte aligned with lda/ldw %% 64 == 0");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256e_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES + G2_ROWSTAT_BYTES));
        attr_set = true;
    }
    const int tiles_m = M / G2_BM, tiles_n = N / G2_BN;
    hipLaunchKernelGGL((gemm_tn256e_kernel<IS_F16, Epi>), dim3(tiles_m * tiles_n), dim3(G2_THREADS),
                       G2_LDS_BYTES + (epi_row_in<Epi>::value ? G2_ROWSTAT_BYTES : 0), st,
                       A, lda, W, ldw, K, tiles_n, epi, gemm_order2d());
    VQ_HIP(hipGetLastError());
    return 0;
}

template <bool IS_F16, class Epi, bool BUF = true /* false: global_load_lds staging (A/B: 1.5-3.5 % slower, 18 more VGPRs) */>
static int launch_gemm_tn256d(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                              int M, int N, int K, const Epi& epi) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0,
             "gemm_tn256d: shape M=%d N=%d K=%d is not tile-aligned (256/256/128)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn256d: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;       // per instantiation; one device per process (vq_init)
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256d_kernel<IS_F16, Epi, 0, BUF>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES + G2_ROWSTAT_BYTES));
        attr_set = true;
    }
    const int tiles_m = M / G2_BM, tiles_n = N / G2_BN;
    hipLaunchKernelGGL((gemm_tn256d_kernel<IS_F16, Epi, 0, BUF>), dim3(tiles_m * tiles_n), dim3(G2_THREADS),
                       G2_LDS_BYTES + (epi_row_in<Epi>::value ? G2_ROWSTAT_BYTES : 0), st,
                       A, lda, W, ldw, K, tiles_n, epi, gemm_order2d(), (unsigned long long*)nullptr);
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace vq

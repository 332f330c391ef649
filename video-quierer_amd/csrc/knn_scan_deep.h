// Third-generation batch scan: scan2_f16_top2_kernel (knn_scan_f16.h) on the deep-prefetch mainloop of
// gemm_mfma256d.h — same 256 queries x 2048 rows per workgroup, same LDS image, same streams and key layout
// (rescore layout 2), same fold of a finished quadrant in the read half of the following phase.  What changes is the
// staging: units are the LDS rows ONE phase reads (A0/A1 = 64-row halves of each wave's query rows, W0/W1 = 32-row
// halves of each wave's matrix rows), each re-issued 2-3 phases after its last read, i.e. 5-6 phases (1.25 K-tiles)
// before its first read, as `buffer_load ... lds` (one descriptor per operand, a 32-bit lane offset, the K / row-tile
// offset in an SGPR).  See gemm_mfma256d.h for the schedule table and the RAW / WAR argument; the flattened K-tile
// index runs over the 8 row tiles of the workgroup's range, so the DMA stream never stops at a row-tile boundary.
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "knn_scan_f16.h"

namespace vq {

constexpr int SCAN4_LDS_BYTES = G2_LDS_BYTES + 8 * 8 * 64 * 8;      // + the running keys: 160 KiB, the whole LDS of a CU

__global__ __launch_bounds__(G2_THREADS, 2)
void scan4_f16_top2_kernel(const uint16_t* __restrict__ Q16, const uint16_t* __restrict__ X16,
                           int dim, int64_t n_valid, int q_tiles, int n_ranges, int range_groups, int64_t q_pad,
                           uint32_t* __restrict__ keys /*batch_key_index (knn_scan_f16.h)*/) {
    typedef mfma_op<true> op;
    typedef op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // workgroup -> (row range, query tile): 4 ranges x 8 query tiles per 32 consecutive workgroups of an XCD (scan2)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int blk = wg >> 5, inner = wg & 31;
    const int rg = blk % range_groups, qg = blk / range_groups;
    const int range = rg * 4 + (inner >> 3);
    const int qtile = qg * 8 + (inner & 7);
    if (range >= n_ranges || qtile >= q_tiles) return;   // whole workgroup leaves before any barrier
    const int m0 = qtile * SCAN2_QT;
    const int64_t n0 = (int64_t)range * SCAN2_RANGE;

    // ---- LDS-DMA: wave w fills pieces p = 2w, 2w+1 (8 rows x 128 B each) of every staging unit ----
    // A unit hm: piece p -> rows (p>>3)*128 + hm*64 + (p&7)*8 ..+7;  W unit hn: piece p -> rows (p>>2)*64 + hn*32 + (p&3)*8 ..+7.
    // The lane's byte offset inside the tile is kept for piece 2w only (unit 0): piece 2w+1 lies 8 rows further and its
    // swizzle differs by chunk ^ 4 (= byte offset ^ 64, rows being multiples of 128 bytes); the unit's row offset, the
    // 8 rows and the K / row-tile offset all go into the scalar offset — 4 lane-offset registers instead of 8.
    const int srow = lane >> 3, sslot = lane & 7;
    const int arow_w = (wave >> 2) * 128 + (wave & 3) * 16, wrow_w = (wave >> 1) * 64 + (wave & 1) * 16;   // piece 2w of unit 0
    const int ar = arow_w + srow, wrw = wrow_w + srow;
    const int a_v0 = (ar * dim + (sslot ^ ((ar >> 1) & 7)) * 8) * 2, a_v1 = a_v0 ^ 64;
    const int w_v0 = (wrw * dim + (sslot ^ ((wrw >> 1) & 7)) * 8) * 2, w_v1 = w_v0 ^ 64;
    const int a_dst0 = arow_w * 128, w_dst0 = 2 * G2_HALF + wrow_w * 128;                  // + unit*{64,32} rows, + 8 rows for piece 2w+1
    const int row8 = 8 * dim * 2;                                                          // 8 rows of the source, bytes
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(Q16 + (size_t)m0 * dim), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)(X16 + (size_t)n0 * dim), 0, 0x7fffffff, 0x00020000);

    const int nk = dim / G2_BK;                          // K-tiles per row tile
    const int total = 8 * nk;                            // flattened K-tiles
    const int tile_bytes = 256 * dim * 2;                // one row tile of the matrix

    // (t, kk) = (row tile, K-tile inside it) of the flattened K-tile being staged
    auto stage_a = [&](int buf, int hm, int kk) __attribute__((always_inline)) {
        char* base = smem + buf * G2_BUF + a_dst0 + hm * (64 * 128);
        const int soff = __builtin_amdgcn_readfirstlane(kk * (G2_BK * 2) + hm * 8 * row8);      // an SGPR: a VGPR here costs a waterfall loop per load
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base), 16, a_v0, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + 1024), 16, a_v1, soff + row8, 0, 0);
    };
    auto stage_w = [&](int buf, int hn, int t, int kk) __attribute__((always_inline)) {
        char* base = smem + buf * G2_BUF + w_dst0 + hn * (32 * 128);
        const int soff = __builtin_amdgcn_readfirstlane(t * tile_bytes + kk * (G2_BK * 2) + hn * 4 * row8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base), 16, w_v0, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + 1024), 16, w_v1, soff + row8, 0, 0);
    };

    // ---- fragment read offsets (identical to scan2 / gemm_tn256d_kernel) ----
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
    frag af[4][2], wf[2][2][2];          // one query sub-block (64 rows); BOTH matrix sub-blocks (2 x 32 rows)
    const float NEG = -__builtin_inff();
    const float MASKED = -3.0e38f;       // finite: see scan_f16_top2_kernel
    // The running (best, second) keys of a lane's 8 query columns live in the 32 KiB of LDS behind the two K-tile
    // buffers, [wave][column][lane] float2 (a wave-wide ds_read_b64 is 512 contiguous bytes: conflict-free): as 16
    // registers they pushed the kernel over its 256-register budget, and a scratch reload inside the K loop waits
    // vmcnt(0), i.e. for every LDS-DMA unit in flight (measured: 16.2 ms against 12.7 ms for the four-phase scan).
    float2* mm = (float2*)(smem + G2_LDS_BYTES) + wave * (8 * 64) + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm[i * 64] = float2{NEG, NEG};

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
    // fold quadrant (hm, hn) of row tile t into the running top-2 and clear it.  Four vector instructions per score (mask the
    // index bits in, v_med3_f32 for the second key, a raw v_max_f32 for the first — fmaxf, and med3(first, key, +inf) which the compiler turns
    // back into it, cost an extra canonicalising v_max per call in IEEE mode) plus the clear.  The row mask of a ragged last tile is a separate copy of the
    // loop behind a wave-uniform branch: written as `if (ragged && row >= n_valid)` the compiler if-converted it into a 64-bit
    // add, a 64-bit compare and two selects PER SCORE on every tile (11 instructions per score; the fold was a quarter of the
    // kernel).
    auto fold_body = [&](auto ragged_tag, int hm, int hn, int t) __attribute__((always_inline)) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        float2 p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = mm[(hm * 4 + i) * 64];
        const int row_lane = t * 256 + wc * 64 + 4 * fgrp;              // + ni*16 + r: the row inside the range
        const int rows_left = (int)min((int64_t)SCAN2_RANGE, n_valid - n0);   // RAGGED only
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mi = hm * 4 + i, ni = hn * 2 + j;
                    float v = acc[mi][ni][r];
                    if constexpr (RAGGED) { if (row_lane + ni * 16 + r >= rows_left) v = MASKED; }
                    const uint32_t kb = (__builtin_bit_cast(uint32_t, v) & ~127u) | (uint32_t)(t * 16 + ni * 4 + r);
                    const float kf = __builtin_bit_cast(float, kb);
                    p[i].y = __builtin_amdgcn_fmed3f(p[i].x, p[i].y, kf);
                    asm("v_max_f32 %0, %1, %2" : "=v"(p[i].x) : "v"(p[i].x), "v"(kf));      // raw: no canonicalising pre-max
                    acc[mi][ni][r] = 0.f;
                }
            mm[(hm * 4 + i) * 64] = p[i];
        }
    };
    auto fold = [&](int hm, int hn, int t) __attribute__((always_inline)) {
        const bool ragged = n0 + (int64_t)(t + 1) * 256 > n_valid;     // wave-uniform; true only in the matrix's last range
        if (__builtin_expect(ragged, 0)) fold_body(std::true_type{}, hm, hn, t);
        else                             fold_body(std::false_type{}, hm, hn, t);
    };
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#define VQ_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

    // One K-tile: the four phases, staging schedule and counted waits of gemm_tn256d_kernel.  `last` = final K-tile of a
    // row tile: each quadrant is folded in the read half of the phase after its last MFMAs; the fourth quadrant's fold
    // lands in phase 1 of the next K-tile (`fold_prev`).
    auto tile = [&](int kt, int bufi, bool last, bool fold_prev, int t, int kk) __attribute__((always_inline)) {
        const char* buf = smem + bufi * G2_BUF;
        const bool next = kt + 1 < total, next2 = kt + 2 < total;
        const int kk1 = kk + 1 == nk ? 0 : kk + 1, t1 = kk + 1 == nk ? t + 1 : t;          // K-tile kt + 1
        const int kk2 = kk1 + 1 == nk ? 0 : kk1 + 1, t2 = kk1 + 1 == nk ? t1 + 1 : t1;     // K-tile kt + 2
        // Where the fold runs (s_memtime stamps): in the READ half of a phase it sits beside the partner wave's
        // MFMA cluster, which holds the SIMD's vector issue for half of every 16 cycles — a quadrant's 96 vector instructions
        // took ~1,000 cycles there plus ~340 waiting for the running keys, and a row tile cost 42.5k cycles against 30.5k with no
        // fold at all.  In the wave's own MFMA half the partner is reading, not multiplying, and the vector pipe is free once
        // the 16 MFMAs are issued.  Two clusters have registers to spare (the second W sub-block is dead in phases 4 and 1):
        // quadrants (0,0) (0,1) are folded behind the MFMAs of phase 4 of a row tile's last K-tile, (1,1) (1,0) behind those
        // of phase 1 of the next K-tile — each before its accumulators are written again.
        // phase 1: quadrant (0,0)
        load_a(buf, 0); load_w(buf, 0);
        if (next) { stage_w(bufi ^ 1, 1, t1, kk1); VQ_VMCNT(8); }
        else      { VQ_VMCNT(2); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 0);
        if (fold_prev) { fold(1, 1, t - 1); fold(1, 0, t - 1); }
        barrier();
        // phase 2: quadrant (0,1)
        load_w(buf, 1);
        if (next) { stage_a(bufi ^ 1, 1, kk1); VQ_VMCNT(8); }
        else      { VQ_VMCNT(0); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 1);
        barrier();
        // phase 3: quadrant (1,1)
        load_a(buf, 1);
        if (next2) stage_a(bufi, 0, kk2);
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 1);
        barrier();
        // phase 4: quadrant (1,0): no fragment reads
        if (next2)     { stage_w(bufi, 0, t2, kk2); VQ_VMCNT(8); }
        else if (next) { VQ_VMCNT(4); }
        barrier();
        mfma_quadrant(1, 0);
        if (last) { fold(0, 0, t); fold(0, 1, t); }
        barrier();
    };

    // ---- prologue: tile 0 complete + A0, W0 of tile 1 in flight; A0(0), W0(0) landed ----
    stage_a(0, 0, 0); stage_w(0, 0, 0, 0); stage_w(0, 1, 0, 0); stage_a(0, 1, 0);
    stage_a(1, 0, 1); stage_w(1, 0, 0, 1);               // nk >= 2: K-tile 1 is still in row tile 0
    VQ_VMCNT(8);
    barrier();

    if (wr == 1) barrier();               // stagger: group 1 runs one barrier behind group 0
    int kk = 0, t = 0;
    for (int kt = 0; kt < total; kt += 2) {              // nk is even: a row tile never ends on an odd kt
        tile(kt, 0, false, kk == 0 && t > 0, t, kk);
        ++kk;
        tile(kt + 1, 1, kk + 1 == nk, false, t, kk);
        if (++kk == nk) { kk = 0; ++t; }
    }
    fold(1, 1, 7); fold(1, 0, 7);
    if (wr == 0) barrier();               // every wave executes the same number of barriers
#undef VQ_VMCNT

    const int64_t stream = (int64_t)range * 16 + wc * 4 + fgrp;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int q = m0 + wr * 128 + mi * 16 + frow;
        const float2 p = mm[mi * 64];
        *(uint2*)(keys + batch_key_index(stream, q, (int64_t)n_ranges * 16)) = uint2{__builtin_bit_cast(uint32_t, p.x), __builtin_bit_cast(uint32_t, p.y)};
    }
}

}  // namespace vq

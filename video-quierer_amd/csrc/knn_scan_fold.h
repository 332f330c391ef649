// Fourth-generation batch scan: scan4_f16_top2_kernel (knn_scan_deep.h) with the top-2 fold taken OUT of the row-tile
// boundary and spread behind the MFMA clusters that follow it.
//
// What the fold costs (measured, profiles/r03_scan_fold_ab.txt — NOT the "third of the kernel" round 2 inferred from two
// geometries): scan4, which folds in two bursts per row tile (behind phase 4 of its last K-tile and phase 1 of the next one,
// 640 vector instructions per lane and row tile: and, or, med3, max, clear per score), 9.79-9.83 ms; this mainloop with no fold
// in the loop 9.12 (DIAG = 1); with the fold after each cluster, not interleaved, 9.69 (DIAG = 2); as built here 9.53-9.55.
//
// Here, per score: v_and_or_b32 (index bits in, mask in a VGPR so the index can be the one scalar operand), v_med3_f32 (second
// key), v_med3_f32 against a large finite constant (first key: a max the compiler cannot fold back into the canonicalising
// form) = 3 instructions, and no clear: the first K-tile of a row tile starts its accumulators from the MFMA's C = 0 operand
// (template ZEROC; with the clear kept, 4 per score, the kernel measures the same: 9.37-9.40 against 9.36-9.41 ms).  A finished
// quadrant's accumulators stay untouched until that quadrant's own cluster of the next row tile's first K-tile, i.e. for the
// three clusters in between; its 96 fold instructions are dealt over those clusters in units of one 16-row block (24
// instructions), 2 or 3 blocks per cluster, and placed BY HAND behind the cluster's 16 MFMAs: after each MFMA one or two scores,
// pinned by sched_barrier(0) — an MFMA holds the SIMD's vector issue for 8 of its 16 cycles (MI355X_MICROARCH.md, cycle
// constants), so 3-6 four-cycle instructions per MFMA stretch such a cluster a little, on six of a row tile's 32 clusters.
// (sched_group_barrier kept all three blocks' temporaries live across the cluster: 59 spills.)  Every K-tile variant — first /
// inside / last of a row tile — is its own straight-line body: conditional copies of a cluster meet in phi nodes over the 32
// accumulators they write and the allocator then spills INTO the K loop.  Row tile 0's first K-tile "folds" accumulators that
// were pre-set to the masked sentinel, so there is no t == 0 copy either.
//
//   cluster (phase)            multiplies     folds (quadrant: 16-row blocks)
//   last K-tile, phase 2       (0,1)          (0,0): 0 1 2
//   last K-tile, phase 3       (1,1)          (0,0): 3      (0,1): 0 1
//   last K-tile, phase 4       (1,0)          (0,1): 2 3
//   first K-tile, phase 1      (0,0)          (1,1): 0 1 2
//   first K-tile, phase 2      (0,1)          (1,1): 3      (1,0): 0 1
//   first K-tile, phase 3      (1,1)          (1,0): 2 3
//   first K-tile, phase 4      (1,0)          -
// Every fold reads a quadrant after its last MFMA of the row tile and before its first MFMA of the next (program order of
// one wave: no cross-wave hazard).  The last row tile of the range is folded after the loop.  Ragged tiles (the matrix's last
// range only) take the same schedule with the row mask compiled in, behind a wave-uniform branch.
//
// Staging, LDS image, key layout, streams: scan4's.  Results are bit-identical to scan4's keys.
#pragma once
#ifndef VQ_SCAN_NT_KEYS
#define VQ_SCAN_NT_KEYS 0
#endif
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "knn_scan_f16.h"
#include "knn_scan_deep.h"

namespace vq {

// fold blocks dealt to one cluster: N of {query column mi (0..7), matrix-column half hn (0..1)}
template <int N_, int M0 = 0, int H0 = 0, int M1 = 0, int H1 = 0, int M2 = 0, int H2 = 0>
struct FoldSpec {
    static constexpr int N = N_;
    static constexpr int mi(int s) { return s == 0 ? M0 : s == 1 ? M1 : M2; }
    static constexpr int hn(int s) { return s == 0 ? H0 : s == 1 ? H1 : H2; }
};

template <int DIAG /* 0 = product; diagnostic builds: 1 = no fold at all (keys invalid), 2 = fold not interleaved (after the cluster) */,
          bool ZEROC = true /* a row tile's first K-tile multiplies into the MFMA's C = 0 operand and the fold does not clear */>
__global__ __launch_bounds__(G2_THREADS, 2)
void scan5_f16_top2_kernel(const uint16_t* __restrict__ Q16, const uint16_t* __restrict__ X16,
                           int dim, int64_t n_valid, int q_tiles, int n_ranges, int range_groups, int64_t q_pad,
                           uint32_t* __restrict__ keys /*batch_key_index (knn_scan_f16.h)*/, int ldx /* row stride of X16, elements */,
                           int rb_log2 = 2 /* 32 consecutive workgroups of an XCD = 2^rb_log2 row ranges x 32 / 2^rb_log2 query tiles */) {
    typedef mfma_op<true> op;
    typedef op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int blk = wg >> 5, inner = wg & 31;
    const int rg = blk % range_groups, qg = blk / range_groups;
    const int qb_log2 = 5 - rb_log2;
    const int range = (rg << rb_log2) + (inner >> qb_log2);
    const int qtile = (qg << qb_log2) + (inner & ((1 << qb_log2) - 1));
    if (range >= n_ranges || qtile >= q_tiles) return;   // whole workgroup leaves before any barrier
    const int m0 = qtile * SCAN2_QT;
    const int64_t n0 = (int64_t)range * SCAN2_RANGE;

    // ---- LDS-DMA (scan4): 4 lane-offset registers, unit / piece / K / row-tile offsets scalar ----
    const int srow = lane >> 3, sslot = lane & 7;
    const int arow_w = (wave >> 2) * 128 + (wave & 3) * 16, wrow_w = (wave >> 1) * 64 + (wave & 1) * 16;
    const int ar = arow_w + srow, wrw = wrow_w + srow;
    const int a_v0 = (ar * dim + (sslot ^ ((ar >> 1) & 7)) * 8) * 2, a_v1 = a_v0 ^ 64;
    const int w_v0 = (wrw * ldx + (sslot ^ ((wrw >> 1) & 7)) * 8) * 2, w_v1 = w_v0 ^ 64;
    const int a_dst0 = arow_w * 128, w_dst0 = 2 * G2_HALF + wrow_w * 128;
    const int row8 = 8 * dim * 2, xrow8 = 8 * ldx * 2;
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(Q16 + (size_t)m0 * dim), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)(X16 + (size_t)n0 * ldx), 0, 0x7fffffff, 0x00020000);

    const int nk = dim / G2_BK;                          // K-tiles per row tile (even)
    const int total = 8 * nk;
    const int tile_bytes = 256 * ldx * 2;

    auto stage_a = [&](int buf, int hm, int kk) __attribute__((always_inline)) {
        char* base = smem + buf * G2_BUF + a_dst0 + hm * (64 * 128);
        const int soff = __builtin_amdgcn_readfirstlane(kk * (G2_BK * 2) + hm * 8 * row8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base), 16, a_v0, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + 1024), 16, a_v1, soff + row8, 0, 0);
    };
    auto stage_w = [&](int buf, int hn, int t, int kk) __attribute__((always_inline)) {
        char* base = smem + buf * G2_BUF + w_dst0 + hn * (32 * 128);
        const int soff = __builtin_amdgcn_readfirstlane(t * tile_bytes + kk * (G2_BK * 2) + hn * 4 * xrow8);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base), 16, w_v0, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + 1024), 16, w_v1, soff + xrow8, 0, 0);
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int fx = (frow >> 1) & 7;
    const int slot[2] = {((0 + fgrp) ^ fx) * 16, ((4 + fgrp) ^ fx) * 16};
    const int a_base = wr * G2_HALF + frow * 128;
    const int w_base = 2 * G2_HALF + (wc >> 1) * G2_HALF + ((wc & 1) * 64 + frow) * 128;

    const float NEG = -__builtin_inff();
    const float MASKED = -3.0e38f;       // finite: see scan_f16_top2_kernel
    f32x4 acc[8][4];                     // rows 64..127 of the wave's queries start at MASKED: see the main loop
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float z = (DIAG != 1 && i >= 4) ? MASKED : 0.f; acc[i][j] = f32x4{z, z, z, z}; }
    frag af[4][2], wf[2][2][2];
    // first-key update as med3(first, key, BIG): = max(first, key) for every key below BIG, and not foldable into fmaxf (whose
    // IEEE-mode lowering canonicalises both operands first: two extra instructions per score)
    float BIG = 3.0e38f;
    uint32_t keep_mask = ~127u;          // in a VGPR: v_and_or_b32 then takes (score, mask, index) with the index as its one scalar
    asm volatile("" : "+v"(BIG), "+v"(keep_mask));
    float2* mm = (float2*)(smem + G2_LDS_BYTES) + wave * (8 * 64) + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) mm[i * 64] = float2{NEG, NEG};
    // ragged = some row of this range lies beyond the matrix: only the matrix's last range (wave-uniform)
    const bool ragged = n0 + SCAN2_RANGE > n_valid;

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
    // one score of a finished row tile -> the running top-2 of its query column; clears it.  e = j*4 + r inside the block
    // (query column mi, matrix-row half hn)
    auto fold_score = [&](auto ragged_tag, int mi, int hn, int e, int t, float2& p) __attribute__((always_inline)) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const int ni = hn * 2 + (e >> 2), r = e & 3;
        float v = acc[mi][ni][r];
        if constexpr (RAGGED) {
            const int rows_left = (int)min((int64_t)SCAN2_RANGE, n_valid - n0);
            if (t * 256 + wc * 64 + 4 * fgrp + ni * 16 + r >= rows_left) v = MASKED;
        }
        const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((t * 16 + ni * 4 + r) & 127);     // & 127: t = -1 on row tile 0's first K-tile
        const float kf = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, v) & keep_mask) | idx);
        p.y = __builtin_amdgcn_fmed3f(p.x, p.y, kf);
        p.x = __builtin_amdgcn_fmed3f(p.x, kf, BIG);
        if constexpr (!ZEROC) acc[mi][ni][r] = 0.f;
    };
    auto fold_block = [&](auto ragged_tag, int mi, int hn, int t, float2& p) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) fold_score(ragged_tag, mi, hn, e, t, p);
    };
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
#define VQ_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

    // one MFMA of a cluster (m = ks*8 + i*2 + j); `fresh`: the first K-tile of a row tile starts from C = 0
    auto mfma1 = [&](auto fresh_tag, int hm, int hn, int m) __attribute__((always_inline)) {
        const int ks = m >> 3, i = (m >> 1) & 3, j = m & 1;
        f32x4& c = acc[hm * 4 + i][hn * 2 + j];
        if (decltype(fresh_tag)::value && ks == 0) c = op::run(wf[hn][j][ks], af[i][ks], f32x4{0.f, 0.f, 0.f, 0.f});
        else                                       c = op::run(wf[hn][j][ks], af[i][ks], c);
    };
    auto mfmas = [&](auto fresh_tag, int hm, int hn) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 16; ++m) mfma1(fresh_tag, hm, hn, m);
    };
    // 16 MFMAs of quadrant (hm, hn) with the fold blocks of `spec` (row tile ft) dealt between them BY HAND: after MFMA m come
    // the scores [m * Q / 16, (m + 1) * Q / 16) of the cluster's Q = 8 N (one or two: 4 or 8 vector instructions in a 16-cycle
    // MFMA shadow of which the MFMA itself holds the issue port for 8), pinned by sched_barrier(0) on both sides.  One block's
    // running pair is live at a time: it is read from LDS one MFMA before its first score and written back after its last.
    // (Left to sched_group_barrier the same work kept all three pairs and their temporaries live across the cluster: 59 spills.)
    auto cluster_fold = [&](auto fresh_tag, auto ragged_tag, int hm, int hn, auto spec, int ft) __attribute__((always_inline)) {
        typedef decltype(spec) S;
        constexpr int Q = S::N * 8;
        float2 p[3];
        p[0] = mm[S::mi(0) * 64];
        if constexpr (DIAG == 2) {
            mfmas(fresh_tag, hm, hn);
#pragma unroll
            for (int s = 0; s < 3; ++s) if (s < S::N) { if (s) p[s] = mm[S::mi(s) * 64]; fold_block(ragged_tag, S::mi(s), S::hn(s), ft, p[s]); mm[S::mi(s) * 64] = p[s]; }
            return;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            mfma1(fresh_tag, hm, hn, m);
            __builtin_amdgcn_sched_barrier(0);
            const int q0 = m * Q / 16, q1 = (m + 1) * Q / 16;
            // the pair of the block that starts in the NEXT gap
            const int qn = (m + 2) * Q / 16 < Q ? (m + 2) * Q / 16 : Q;
#pragma unroll
            for (int q = q1; q < qn; ++q) if ((q & 7) == 0 && (q >> 3) > 0 && (q >> 3) < S::N) p[q >> 3] = mm[S::mi(q >> 3) * 64];
#pragma unroll
            for (int q = q0; q < q1; ++q) {
                const int sblk = q >> 3, e = q & 7;
                fold_score(ragged_tag, S::mi(sblk), S::hn(sblk), e, ft, p[sblk]);
                if (e == 7) mm[S::mi(sblk) * 64] = p[sblk];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto cluster = [&](auto fresh_tag, auto ragged_tag, int hm, int hn, auto spec, int ft) __attribute__((always_inline)) {
        typedef decltype(spec) S;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (DIAG == 1 || S::N == 0) mfmas(fresh_tag, hm, hn);
        else cluster_fold(fresh_tag, ragged_tag, hm, hn, spec, ft);
        __builtin_amdgcn_s_setprio(0);
    };

    // One K-tile.  KIND 0: inside a row tile (no fold); 1: first K-tile of a row tile (carries the second half of the previous row
    // tile's fold — for t = 0 that folds the kernel's initial accumulators, see below); 2: last K-tile of a row tile (first half of
    // this tile's fold).  Every variant is straight-line code around its MFMAs: conditional copies of a cluster meet in phi
    // nodes over the 32 accumulators they write, and the register allocator then spills (measured: 15-77 registers).
    auto tile = [&](auto kind_tag, auto ragged_tag, int kt, int bufi, int t, int kk) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_tag)::value;
        const std::integral_constant<bool, KIND == 1 && ZEROC && DIAG != 1> fresh{};
        const char* buf = smem + bufi * G2_BUF;
        const bool next = kt + 1 < total, next2 = kt + 2 < total;
        const int kk1 = kk + 1 == nk ? 0 : kk + 1, t1 = kk + 1 == nk ? t + 1 : t;
        const int kk2 = kk1 + 1 == nk ? 0 : kk1 + 1, t2 = kk1 + 1 == nk ? t1 + 1 : t1;
        // phase 1: quadrant (0,0)
        load_a(buf, 0); load_w(buf, 0);
        if (next) { stage_w(bufi ^ 1, 1, t1, kk1); VQ_VMCNT(8); }
        else      { VQ_VMCNT(2); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KIND == 1) cluster(fresh, ragged_tag, 0, 0, FoldSpec<3, 4, 1, 5, 1, 6, 1>{}, t - 1);       // (1,1): 0 1 2
        else                     cluster(fresh, ragged_tag, 0, 0, FoldSpec<0>{}, t);
        barrier();
        // phase 2: quadrant (0,1)
        load_w(buf, 1);
        if (next) { stage_a(bufi ^ 1, 1, kk1); VQ_VMCNT(8); }
        else      { VQ_VMCNT(0); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KIND == 2)      cluster(fresh, ragged_tag, 0, 1, FoldSpec<3, 0, 0, 1, 0, 2, 0>{}, t);      // (0,0): 0 1 2
        else if constexpr (KIND == 1) cluster(fresh, ragged_tag, 0, 1, FoldSpec<3, 7, 1, 4, 0, 5, 0>{}, t - 1);  // (1,1): 3   (1,0): 0 1
        else                          cluster(fresh, ragged_tag, 0, 1, FoldSpec<0>{}, t);
        barrier();
        // phase 3: quadrant (1,1)
        load_a(buf, 1);
        if (next2) stage_a(bufi, 0, kk2);
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KIND == 2)      cluster(fresh, ragged_tag, 1, 1, FoldSpec<3, 3, 0, 0, 1, 1, 1>{}, t);      // (0,0): 3   (0,1): 0 1
        else if constexpr (KIND == 1) cluster(fresh, ragged_tag, 1, 1, FoldSpec<2, 6, 0, 7, 0>{}, t - 1);        // (1,0): 2 3
        else                          cluster(fresh, ragged_tag, 1, 1, FoldSpec<0>{}, t);
        barrier();
        // phase 4: quadrant (1,0): no fragment reads
        if (next2)     { stage_w(bufi, 0, t2, kk2); VQ_VMCNT(8); }
        else if (next) { VQ_VMCNT(4); }
        barrier();
        if constexpr (KIND == 2) cluster(fresh, ragged_tag, 1, 0, FoldSpec<2, 2, 1, 3, 1>{}, t);                 // (0,1): 2 3
        else                     cluster(fresh, ragged_tag, 1, 0, FoldSpec<0>{}, t);
        barrier();
    };

    // ---- prologue: tile 0 complete + A0, W0 of tile 1 in flight; A0(0), W0(0) landed ----
    stage_a(0, 0, 0); stage_w(0, 0, 0, 0); stage_w(0, 1, 0, 0); stage_a(0, 1, 0);
    stage_a(1, 0, 1); stage_w(1, 0, 0, 1);               // nk >= 2: K-tile 1 is still in row tile 0
    VQ_VMCNT(8);
    barrier();

    if (wr == 1) barrier();               // stagger: group 1 runs one barrier behind group 0
    const std::true_type T{}; const std::false_type F{};
    const std::integral_constant<int, 0> MID{}; const std::integral_constant<int, 1> FIRST{}; const std::integral_constant<int, 2> LAST{};
    // Row tile 0's first K-tile also "folds the previous row tile's" quadrants (1,1) (1,0): those accumulators start at MASKED
    // instead of 0, so what enters the running keys there loses to every real score (each stream sees 128 of them) and the
    // fold's clear leaves the accumulators at 0 before their first MFMA — no t == 0 copy of the K-tile body.
    auto run = [&](auto ragged_tag) __attribute__((always_inline)) {
        int kt = 0;
        for (int t = 0; t < 8; ++t) {
            tile(FIRST, ragged_tag, kt, 0, t, 0); ++kt;
            for (int kk = 1; kk + 1 < nk; kk += 2) {     // nk even: the K-tiles between first and last come in pairs
                tile(MID, ragged_tag, kt, 1, t, kk); ++kt;
                tile(MID, ragged_tag, kt, 0, t, kk + 1); ++kt;
            }
            tile(LAST, ragged_tag, kt, 1, t, nk - 1); ++kt;
        }
        // the last row tile's second half: nothing left to hide it behind
        if constexpr (DIAG != 1) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {                // quadrants (1,1) then (1,0)
                const int hn = q == 0 ? 1 : 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float2 p = mm[(4 + i) * 64];
                    fold_block(ragged_tag, 4 + i, hn, 7, p);
                    mm[(4 + i) * 64] = p;
                }
            }
        } else {                                         // diagnostic: one fold at the very end keeps the MFMAs alive
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                float2 p = mm[mi * 64];
                fold_block(ragged_tag, mi, 0, 7, p); fold_block(ragged_tag, mi, 1, 7, p);
                mm[mi * 64] = p;
            }
        }
    };
    if (__builtin_expect(ragged, 0)) run(T); else run(F);
    if (wr == 0) barrier();               // every wave executes the same number of barriers
#undef VQ_VMCNT

    const int64_t stream = (int64_t)range * 16 + wc * 4 + fgrp;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int q = m0 + wr * 128 + mi * 16 + frow;
        const float2 p = mm[mi * 64];
#if VQ_SCAN_NT_KEYS      // the keys (625 MB per 10k x 1M batch, read once by the re-score kernel) as streaming stores: build-time A/B
        { typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
          __builtin_nontemporal_store(u32x2{__builtin_bit_cast(uint32_t, p.x), __builtin_bit_cast(uint32_t, p.y)}, (u32x2*)(keys + batch_key_index(stream, q, (int64_t)n_ranges * 16))); }
#else
        *(uint2*)(keys + batch_key_index(stream, q, (int64_t)n_ranges * 16)) = uint2{__builtin_bit_cast(uint32_t, p.x), __builtin_bit_cast(uint32_t, p.y)};
#endif
    }
}

}  // namespace vq

// fp16 MFMA scan of the embedding matrix with a fused per-stream top-2, and the
// exact re-score / verification pass that turns its output into the exact
// answer (SURVEY.md §7 step 3 "exactness guard", §8a K6).
//
// Pass 1  scan_f16_top2_kernel
//   scores = Q16 (queries, fp16) x X16^T (matrix rows, fp16), fp32 accumulate, never
//   materialised: every lane keeps, for each of its 4 query columns, the two
//   largest scores of its own "stream" of 128 matrix rows and writes them once.
//   A workgroup covers 128 queries x 1024 rows (8 row tiles, one continuous
//   software-pipelined k-loop); stream = (1024-row range, wave column wn, lane
//   group g) = rows  range*1024 + t*128 + wn*64 + ni*16 + 4g + r,  local index
//   t*16 + ni*4 + r (7 bits) stored in the low mantissa bits of the score.
//   Output: keys[stream][query][2] (fp32 bit patterns), 8 B per (stream, query).
//
// Pass 2  rescore_verify_kernel  (RV_QPW = 8 queries per workgroup)
//   per query: keep the best 4 keys of each of 32 interleaved stream shares,
//   take the best C=32 of those 128, re-score them EXACTLY from the fp32 master
//   (fixed-order fp64 chain, bit-identical to oracle/knn_oracle.c), and prove the
//   result: with s_k the k-th best exact score and EPS the worst-case fp16
//   scoring error, every row that was NOT re-scored has an upper bound
//     - rows a stream did not emit        <= that stream's 2nd key
//     - keys a share dropped              <= the largest key that share dropped
//     - kept keys outside the best C      <= the (C+1)-th kept key
//   If a stream's 2nd key could still matter (key + EPS >= s_k) its 128 rows are
//   re-scored exactly too (up to RESCAN_MAX streams).  If a bound cannot be
//   closed the query is flagged and the caller runs the full exact scan for it.
//   Outcome per query: 0 proven exact, 1 proven exact after stream rescans,
//   2 needs the exact fallback.
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "knn_kernels.h"
#include "knn_scan_small.h"

namespace vq {

constexpr int SCAN_QT = 128;          // queries per workgroup
constexpr int SCAN_RANGE = 1024;      // matrix rows per workgroup (8 tiles of 128)
constexpr int SCAN_STREAM_ROWS = 128; // rows seen by one lane stream
// Worst-case |fp16 score - exact score| for a UNIT row and a UNIT query of dimension dim; it scales with |row||q|:
//   2*2^-11 + 2^-22   fp16 rounding of both operands (Cauchy-Schwarz)
//   dim * 2^-23       fp32 accumulation of the exact products, any order, truncating adders allowed
//   2^-16             index bits packed into the low mantissa bits of the key
//   2 * sqrt(dim) * 2^-25   elements below the fp16 normal range (absolute, not relative, rounding)
// plus 2 % for the fp32 arithmetic that evaluates the bound itself.  1.07e-3 at dim 512, 1.5e-3 at dim 4096.
// The bound above is RELATIVE to |row| |q| except for its last term, and the query is converted to fp16 as given: far below
// unit norm its elements land in fp16's subnormal / flush range, where the rounding error is absolute and does not shrink
// with |q| (at |q| ~ 1e-5 it already exceeds the scaled bound); far above, elements overflow to inf and the packed keys
// become NaN, which v_max drops.  Queries with |q|^2 outside [0.25, 4] (or not finite) are therefore never "proven": the
// re-score kernels file them as state 2 and the exact scan answers them.  The Python hosts normalise their queries
// (hnsw.py:250), so only raw C-ABI callers meet this.
constexpr float SCAN_Q2_MIN = 0.25f, SCAN_Q2_MAX = 4.0f;

static inline float scan_eps_unit(int dim) {
    const double e = 2.0 / 2048 + 1.0 / 4194304 + dim / 8388608.0 + 1.0 / 65536 + 2.0 * __builtin_sqrt((double)dim) / 33554432.0;
    return (float)(e * 1.02);
}

// Key layout of the batch scans: [group of 16 queries][stream][16 queries][2 keys].  A scan lane group writes the
// 128 bytes of (stream, 16 queries) at once, and the re-score workgroup of a query group walks its streams through
// CONTIGUOUS memory (stream-major [stream][q_pad] made every one of its reads a 64-128 byte piece of a different
// 80 KB row: the pass moved 0.67 GB at ~0.7 TB/s).  The streaming scan for <= SCAN3_MAX_Q queries writes [q_pad][stream] (knn_scan_small.h).
__host__ __device__ inline size_t batch_key_index(int64_t stream, int64_t q, int64_t streams) {
    return (((size_t)(q >> 4) * streams + stream) * 16 + (q & 15)) * 2;
}

// row number of (stream, local index)
__host__ __device__ inline int64_t scan_row_of(int64_t stream, int local) {
    const int64_t range = stream >> 3;
    const int wn = (int)(stream >> 2) & 1, g = (int)stream & 3;
    const int t = local >> 4, ni = (local >> 2) & 3, r = local & 3;
    return range * SCAN_RANGE + t * 128 + wn * 64 + ni * 16 + 4 * g + r;
}

__global__ __launch_bounds__(GEMM_THREADS, 2)
void scan_f16_top2_kernel(const uint16_t* __restrict__ Q16, const uint16_t* __restrict__ X16,
                          int dim, int64_t n_valid, int q_tiles, int64_t q_pad,
                          uint32_t* __restrict__ keys /*batch_key_index*/) {
    typedef mfma_op<true> op;
    typedef op::frag frag;
    __shared__ __attribute__((aligned(16))) char smem[GEMM_LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int range = wg / q_tiles;                 // query tile fastest: workgroups that share an XCD's L2
    const int m0 = (wg - range * q_tiles) * SCAN_QT;  //   walk the same 1024 matrix rows
    const int64_t n0 = (int64_t)range * SCAN_RANGE;

    const int srow = lane >> 3, sslot = lane & 7;
    const uint16_t* q_src[4];
    const uint16_t* x_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + srow;
        const int chunk = sslot ^ ((row >> 1) & 7);
        q_src[i] = Q16 + (size_t)(m0 + row) * dim + chunk * 8;
        x_src[i] = X16 + (size_t)(n0 + row) * dim + chunk * 8;
    }
    constexpr int TILE_BYTES = GEMM_BM * GEMM_BK * 2;
    constexpr int BUF_BYTES = 2 * TILE_BYTES;
    const int nk = dim / GEMM_BK;
    const int steps = 8 * nk;

    auto stage = [&](int buf, int s) {
        const int t = s / nk, kt = s - t * nk;
        char* base = smem + buf * BUF_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(q_src[i] + kt * GEMM_BK),
                                             (lds_void_t*)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(x_src[i] + (size_t)t * 128 * dim + kt * GEMM_BK),
                                             (lds_void_t*)(base + TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    const int frow = lane & 15, fgrp = lane >> 4;
    const int fx = (frow >> 1) & 7;
    const int a_off = (wm * 64 + frow) * 128;
    const int w_off = TILE_BYTES + (wn * 64 + frow) * 128;
    const int slot0 = ((0 + fgrp) ^ fx) * 16, slot1 = ((4 + fgrp) ^ fx) * 16;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float NEG = -__builtin_inff();
    // pad rows get a FINITE sentinel: -inf with index bits packed into its mantissa would be a signalling
    // NaN, and v_max_f32 in IEEE mode turns (x, sNaN) into NaN, wiping the running maximum
    const float MASKED = -3.0e38f;
    float m1[4] = {NEG, NEG, NEG, NEG}, m2[4] = {NEG, NEG, NEG, NEG};

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int kt = 0, t = 0;
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        if (s + 1 < steps) stage(cur ^ 1, s + 1);
        const char* buf = smem + cur * BUF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = ks ? slot1 : slot0;
            frag qf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qf[i] = *(const frag*)(buf + a_off + i * 2048 + so);
                xf[i] = *(const frag*)(buf + w_off + i * 2048 + so);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = op::run(xf[ni], qf[mi], acc[mi][ni]);   // rows on the register axis, queries on lanes
        }
        if (++kt == nk) {
            // tile finished: fold its 16 scores per query column into the lane's running top-2
            const bool ragged = n0 + (int64_t)(t + 1) * 128 > n_valid;    // wave-uniform
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[mi][ni][r];
                        if (ragged && n0 + t * 128 + wn * 64 + ni * 16 + 4 * fgrp + r >= n_valid) v = MASKED;
                        const uint32_t kb = (__builtin_bit_cast(uint32_t, v) & ~127u) | (uint32_t)(t * 16 + ni * 4 + r);
                        const float kf = __builtin_bit_cast(float, kb);
                        m2[mi] = __builtin_amdgcn_fmed3f(m1[mi], m2[mi], kf);
                        m1[mi] = fmaxf(m1[mi], kf);
                        acc[mi][ni][r] = 0.f;
                    }
            kt = 0; ++t;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int64_t stream = (int64_t)range * 8 + wn * 4 + fgrp;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int q = m0 + wm * 64 + mi * 16 + frow;
        *(uint2*)(keys + batch_key_index(stream, q, (int64_t)(gridDim.x / q_tiles) * 8)) =
            uint2{__builtin_bit_cast(uint32_t, m1[mi]), __builtin_bit_cast(uint32_t, m2[mi])};
    }
}

// ---------------------------------------------------------------------------------------------
// Second-generation scan: the 256x256 four-phase mainloop of gemm_mfma256.h (staggered wave groups,
// counted vmcnt, raw barriers) with the top-2 fold as its "epilogue".  A workgroup covers 256 queries
// x 2048 matrix rows (8 row tiles, one continuous K loop of 8*dim/64 K-tiles); wave (wr, wc) owns
// queries 128 wr .. +127 and rows 64 wc .. +63 of every tile; stream = (2048-row range, wc, lane
// group g) = rows range*2048 + t*256 + wc*64 + ni*16 + 4g + r, local index t*16 + ni*4 + r.
// The fold of a finished quadrant (32 accumulators) runs in the read half of the following phase,
// i.e. under the partner group's MFMAs.
constexpr int SCAN2_QT = 256;
constexpr int SCAN2_RANGE = 2048;

__host__ __device__ inline int64_t scan2_row_of(int64_t stream, int local) {
    const int64_t range = stream >> 4;
    const int wc = (int)(stream >> 2) & 3, g = (int)stream & 3;
    const int t = local >> 4, ni = (local >> 2) & 3, r = local & 3;
    return range * SCAN2_RANGE + t * 256 + wc * 64 + ni * 16 + 4 * g + r;
}

__global__ __launch_bounds__(G2_THREADS, 2)
void scan2_f16_top2_kernel(const uint16_t* __restrict__ Q16, const uint16_t* __restrict__ X16,
                           int dim, int64_t n_valid, int q_tiles, int n_ranges, int range_groups, int64_t q_pad,
                           uint32_t* __restrict__ keys /*[streams][q_pad][2]*/) {
    typedef mfma_op<true> op;
    typedef op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // Workgroup -> (row range, query tile).  Measured with the plain "query tile fastest" order: 55 % of
    // the L2 requests missed (47 GB from beyond L2 for a 1 GB matrix: every workgroup re-reads its own
    // 256 KB query tile once per row tile and 32 distinct query tiles do not fit a 4 MiB L2), and the
    // DMA ring is too shallow to cover Infinity-Cache latency.  So the 32 workgroups an XCD runs at a
    // time form a 4 (ranges) x 8 (query tiles) block: they walk their ranges in step, which makes the
    // live set 4 row tiles + 8 query tiles = 3 MiB, every row tile is fetched once per 8 workgroups and
    // the 8 query tiles stay L2-resident; blocks advance range-group fastest so those query tiles are
    // reused by the next block too.
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int blk = wg >> 5, inner = wg & 31;
    const int rg = blk % range_groups, qg = blk / range_groups;
    const int range = rg * 4 + (inner >> 3);
    const int qtile = qg * 8 + (inner & 7);
    if (range >= n_ranges || qtile >= q_tiles) return;   // whole workgroup leaves before any barrier
    const int m0 = qtile * SCAN2_QT;
    const int64_t n0 = (int64_t)range * SCAN2_RANGE;

    const int srow = lane >> 3, sslot = lane & 7;
    const uint16_t* a_src[2];
    const uint16_t* w_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + srow;
        const int chunk = sslot ^ ((row >> 1) & 7);
        a_src[i] = Q16 + (size_t)(m0 + row) * dim + chunk * 8;
        w_src[i] = X16 + (size_t)(n0 + row) * dim + chunk * 8;
    }
    const size_t half_rows = (size_t)128 * dim;
    const int piece_off = wave * 2048;
    const int nk = dim / G2_BK;                          // K-tiles per row tile
    const int total = 8 * nk;                            // flattened K-tiles

    // which: 0/1 = query halves, 2/3 = matrix-row halves; kt = flattened K-tile index
    auto stage = [&](int buf, int which, int kt) __attribute__((always_inline)) {
        char* dst = smem + buf * G2_BUF + which * G2_HALF + piece_off;
        const int t = kt / nk, kk = kt - t * nk;
        if (which < 2) {
            const size_t off = (which ? half_rows : 0) + (size_t)kk * G2_BK;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[0] + off), (lds_void_t*)(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[1] + off), (lds_void_t*)(dst + 1024), 16, 0, 0);
        } else {
            const size_t off = (size_t)t * 256 * dim + ((which & 1) ? half_rows : 0) + (size_t)kk * G2_BK;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[0] + off), (lds_void_t*)(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[1] + off), (lds_void_t*)(dst + 1024), 16, 0, 0);
        }
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
    frag af[4][2], wf[2][2];
    const float NEG = -__builtin_inff();
    const float MASKED = -3.0e38f;       // finite: see scan_f16_top2_kernel
    float m1[8], m2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { m1[i] = NEG; m2[i] = NEG; }

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
                wf[j][ks] = *(const frag*)(buf + w_base + (hn * 2 + j) * 2048 + slot[ks]);
    };
    auto mfma_quadrant = [&](int hm, int hn) __attribute__((always_inline)) {
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
    // fold quadrant (hm, hn) of row tile t into the running top-2 and clear it
    auto fold = [&](int hm, int hn, int t) __attribute__((always_inline)) {
        const bool ragged = n0 + (int64_t)(t + 1) * 256 > n_valid;     // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mi = hm * 4 + i, ni = hn * 2 + j;
                    float v = acc[mi][ni][r];
                    if (ragged && n0 + t * 256 + wc * 64 + ni * 16 + 4 * fgrp + r >= n_valid) v = MASKED;
                    const uint32_t kb = (__builtin_bit_cast(uint32_t, v) & ~127u) | (uint32_t)(t * 16 + ni * 4 + r);
                    const float kf = __builtin_bit_cast(float, kb);
                    m2[mi] = __builtin_amdgcn_fmed3f(m1[mi], m2[mi], kf);
                    m1[mi] = fmaxf(m1[mi], kf);
                    acc[mi][ni][r] = 0.f;
                }
    };
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // One K-tile (same four phases, hazards and waits as gemm_tn256_kernel).  `last` = final K-tile of
    // a row tile: each quadrant is folded in the read half of the phase after its last MFMAs; the
    // fourth quadrant's fold lands in phase 1 of the next K-tile (`fold_prev`).
    auto tile = [&](int kt, int bufi, bool last, bool fold_prev, int t) __attribute__((always_inline)) {
        const char* buf = smem + bufi * G2_BUF;
        const bool next = kt + 1 < total, next2 = kt + 2 < total;
        if (fold_prev) fold(1, 0, t - 1);
        load_a(buf, 0); load_w(buf, 0);
        if (next) stage(bufi ^ 1, 1, kt + 1);
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 0);
        barrier();
        if (last) fold(0, 0, t);
        load_w(buf, 1);
        if (next) stage(bufi ^ 1, 2, kt + 1);
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 1);
        barrier();
        if (last) fold(0, 1, t);
        load_a(buf, 1);
        if (next) stage(bufi ^ 1, 3, kt + 1);
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 1);
        barrier();
        if (last) fold(1, 1, t);
        load_w(buf, 0);
        if (next2) { stage(bufi, 0, kt + 2); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        else       { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 0);
        barrier();
    };

    stage(0, 0, 0); stage(0, 1, 0); stage(0, 2, 0); stage(0, 3, 0);
    if (total > 1) { stage(1, 0, 1); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else           { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    barrier();

    if (wr == 1) barrier();
    int kk = 0, t = 0;
    for (int kt = 0; kt < total; kt += 2) {              // nk is even: a row tile never ends on an odd kt
        tile(kt, 0, false, kk == 0 && t > 0, t);
        ++kk;
        tile(kt + 1, 1, kk + 1 == nk, false, t);
        if (++kk == nk) { kk = 0; ++t; }
    }
    fold(1, 0, 7);
    if (wr == 0) barrier();

    const int64_t stream = (int64_t)range * 16 + wc * 4 + fgrp;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int q = m0 + wr * 128 + mi * 16 + frow;
        *(uint2*)(keys + batch_key_index(stream, q, (int64_t)n_ranges * 16)) =
            uint2{__builtin_bit_cast(uint32_t, m1[mi]), __builtin_bit_cast(uint32_t, m2[mi])};
    }
}

// queries fp32 [nq][dim] -> fp16 [q_pad][dim], pad rows zero
__global__ __launch_bounds__(256)
void queries_to_f16_kernel(const float* __restrict__ q, uint16_t* __restrict__ q16, int nq, int64_t q_pad, int dim) {
    const int64_t total4 = q_pad * dim / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = (i * 4) / dim;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < nq) v = *(const float4*)(q + i * 4);
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        const h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *(uint2*)(q16 + i * 4) = __builtin_bit_cast(uint2, h);
    }
}

// ---------------------------------------------------------------------------
constexpr int RV_QPW = 8;         // queries per workgroup (16 needed 90 KB of LDS, mostly the rescan pool: one workgroup per CU; 8 -> three)
constexpr int RV_KEEP = 6;        // keys kept per (query, share).  A wave runs the insertion path whenever ANY of its lanes inserts (most offers), so its
                                  // length sets the cost of phase 1: 8 -> 0.96 ms per 10k queries, 6 -> 0.74, 4 -> 0.71 but then ~3 queries in 10,000 leave
                                  // a bound open (a share holding 5 of the best ~11 keys) and go to the exact fallback
constexpr int RV_SHARES = 256 / RV_QPW;   // stream shares per query (thread = (query, share))
constexpr int RV_TPQ = 256 / RV_QPW;      // threads per query in the selection phases
constexpr int RV_C = 32;          // candidates re-scored exactly
constexpr int RV_RESCAN_MAX = 4;  // streams re-scored per query before giving up
constexpr int RV_POOL = RV_C + RV_RESCAN_MAX * SCAN_STREAM_ROWS;   // exact-scored rows per query

// The same chain with the row streamed in batches of eight 16-byte loads (a row straight from HBM: with one load per step a
// thread paid a full memory round trip per 16 bytes — stamps: 306k of a re-score workgroup's 815k cycles).  dim % 32 == 0.
__device__ __forceinline__ float exact_dot_chain_pf(const float* __restrict__ row, const float* __restrict__ q, int dim) {
    double acc = 0.0;
    for (int i = 0; i < dim; i += 32) {
        float4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *(const float4*)(row + i + 4 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 b = *(const float4*)(q + i + 4 * u);
            acc += (double)a[u].x * (double)b.x;
            acc += (double)a[u].y * (double)b.y;
            acc += (double)a[u].z * (double)b.z;
            acc += (double)a[u].w * (double)b.w;
        }
    }
    return (float)acc;
}

__device__ __forceinline__ float exact_dot_chain(const float* __restrict__ row, const float* __restrict__ q, int dim) {
    double acc = 0.0;
    for (int i = 0; i < dim; i += 4) {
        const float4 a = *(const float4*)(row + i), b = *(const float4*)(q + i);
        acc += (double)a.x * (double)b.x;
        acc += (double)a.y * (double)b.y;
        acc += (double)a.z * (double)b.z;
        acc += (double)a.w * (double)b.w;
    }
    return (float)acc;
}

// keys carry the sign of the score, so compare them as floats; ties and the
// packed low bits do not matter for the proof (only upper bounds are used).
// Template: QPW queries per workgroup, KEEP keys kept per (query, share), C candidates re-scored exactly.  <8, 6, 32> serves
// k <= RV_K_SMALL (what round 2 shipped); <4, 10, 80> serves k up to RV_K_MAX = 64 — the caller over-fetches k * 2
// (video_search_system.py:297), and a proof needs the (C+1)-th key at least a bound's width below the k-th exact score, i.e.
// C well above k: with C = 32, k in (20, 32] sent nearly every query to the exact fallback.  layout 3 = the streaming scan's
// query-major keys (small batches with k > RV_K_SMALL come here instead of rescore_verify_small_kernel).
template <int QPW, int KEEP, int C>
__global__ __launch_bounds__(256)
void rescore_verify_kernel_t(const uint32_t* __restrict__ keys, int64_t streams, int64_t q_pad,
                           const float* __restrict__ rows, int64_t n_valid, int dim,
                           const float* __restrict__ queries, int nq, int k,
                           int32_t* __restrict__ out_ids, float* __restrict__ out_dist,
                           int32_t* __restrict__ flags, int layout /*1: scan_f16_top2, 2: scan2_f16_top2, 3: scan3_f16_top2 streams*/,
                           float eps_rows /* scan_eps_unit(dim) x the largest |row| in the index */,
                           const TieOrder tie /* (distance, id) order of the result: vq_common.h */) {
    constexpr int SHARES = 256 / QPW, TPQ = 256 / QPW, POOL = C + RV_RESCAN_MAX * SCAN_STREAM_ROWS;
    __shared__ float kept_v[QPW][SHARES * KEEP];      // kept key values (with packed index bits)
    __shared__ int kept_s[QPW][SHARES * KEEP];        // stream*2 + which (0: 1st key, 1: 2nd key)
    __shared__ float share_floor[QPW][SHARES];           // upper bound of what a share dropped
    __shared__ int cand_row[QPW][POOL];                  // rows scored exactly
    __shared__ float cand_dist[QPW][POOL];
    __shared__ float cand_key[QPW][C];                   // approx key value of candidate c
    __shared__ int cand_src[QPW][C];
    __shared__ int pool_n[QPW];
    __shared__ float bound_rest[QPW];                       // (C+1)-th kept key
    __shared__ int resc_stream[QPW][RV_RESCAN_MAX];
    __shared__ int resc_n[QPW];
    __shared__ int state[QPW];
    __shared__ float qn2[QPW][SHARES];                   // partial |q|^2 (the bound scales with |q|)
    __shared__ float dk_s[QPW];                             // k-th smallest exact distance among the candidates
    constexpr int LMAX = 4 * C;                                // collected keys per query in step 2
    static_assert(LMAX <= POOL, "the collected list lives in the candidate arrays");
    float (*sel_v)[POOL] = cand_dist;                          // step 2's list: in the candidate arrays, which step 3 writes first
    int (*sel_i)[POOL] = cand_row;                             // (8-10 KB of LDS of its own cost the <8, 6, 32> form a workgroup per CU)
    __shared__ int sel_n[QPW];
    __shared__ float thr1[QPW];

    const int tid = threadIdx.x;
    const int ql = tid % QPW, share = tid / QPW;
    const int q0 = blockIdx.x * QPW;
    const int q = q0 + ql;
    const float NEG = -__builtin_inff();

    // ---- 1. each thread keeps the best KEEP keys of its share (streams s = share mod 16) ----
    float kv[KEEP]; int ksrc[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; ++i) { kv[i] = NEG; ksrc[i] = -1; }
    float dropped = NEG;
    // Insertion by compare-and-swap down the sorted list, as SELECTS: written with `if (v > kv[i]) swap` the compiler built an
    // exec-mask branch per step (~70 instructions per offer, and a wave runs the insertion whenever any of its lanes inserts);
    // raw v_max for the floor (fmaxf costs a canonicalising pre-max per operand in IEEE mode; the keys are never NaN).
    auto raw_max = [](float a, float b) __attribute__((always_inline)) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto offer = [&](float v, int src) __attribute__((always_inline)) {
        if (v > kv[KEEP - 1]) {
#pragma unroll
            for (int i = 0; i < KEEP; ++i) {
                const bool gt = v > kv[i];
                const float hi = gt ? v : kv[i], lo = gt ? kv[i] : v;
                const int shi = gt ? src : ksrc[i], slo = gt ? ksrc[i] : src;
                kv[i] = hi; ksrc[i] = shi; v = lo; src = slo;
            }
        }
        dropped = raw_max(dropped, v);                 // what fell off the list: the old last key, or v itself
    };
    // 8 independent 8-byte loads in flight per thread (a one-load-per-iteration loop was latency-bound:
    // 488 dependent round trips per thread made this kernel 1.8 ms for 10k queries)
    // batches of eight 8-byte loads, the next batch in flight while the current one is offered (one batch at a time waited a
    // full memory round trip per batch: 194k cycles for 32 batches)
    constexpr int PF = QPW >= 8 ? 12 : 8;      // [r03] 12 loads in flight per thread where a thread walks the most keys (61 -> 41 dependent batches at 1M rows)
    auto key_at = [&](int64_t st) __attribute__((always_inline)) {       // element index of stream st's key pair of query q
        return layout == 3 ? ((size_t)q * streams + st) * 2 : batch_key_index(st, q, streams);
    };
    int64_t s = share;
    uint2 two[PF], nxt[PF];
    const bool any_full = s + (PF - 1) * SHARES < streams;
    if (any_full) {
#pragma unroll
        for (int u = 0; u < PF; ++u) two[u] = *(const uint2*)(keys + key_at(s + u * SHARES));   // a wave: 8 adjacent 128-byte blocks
    }
    for (; s + (PF - 1) * SHARES < streams; s += PF * SHARES) {
        const int64_t sn = s + PF * SHARES;
        const bool more = sn + (PF - 1) * SHARES < streams;
        if (more) {
#pragma unroll
            for (int u = 0; u < PF; ++u) nxt[u] = *(const uint2*)(keys + key_at(sn + u * SHARES));
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int src = (int)((s + u * SHARES) * 2);
            offer(__builtin_bit_cast(float, two[u].x), src);
            offer(__builtin_bit_cast(float, two[u].y), src + 1);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < PF; ++u) two[u] = nxt[u];
        }
    }
    for (; s < streams; s += SHARES) {
        const uint2 two = *(const uint2*)(keys + key_at(s));
        offer(__builtin_bit_cast(float, two.x), (int)(s * 2));
        offer(__builtin_bit_cast(float, two.y), (int)(s * 2 + 1));
    }
#pragma unroll
    for (int i = 0; i < KEEP; ++i) { kept_v[ql][share * KEEP + i] = kv[i]; kept_s[ql][share * KEEP + i] = ksrc[i]; }
    share_floor[ql][share] = dropped;
    if (tid < QPW) { pool_n[tid] = 0; resc_n[tid] = 0; state[tid] = 0; }
    __syncthreads();

    // ---- 2. per query (TPQ threads each): the best C of the SHARES * KEEP kept keys and the (C+1)-th as a bound ----
    // C <= 32: rank counting over all kept keys, rank(i) = #{j : v_j > v_i or (v_j == v_i and j < i)} — 1,152 steps per thread
    // at <8, 6, 32>.  At <4, 10, 80> that would be 6,400 (80 us of a single large-k query), so two levels instead, same order,
    // same result:  (a) T1 = the (C+1)-th largest among the FIRST P keys of every share (P = ceil((C+1) / SHARES); a share's
    // list is sorted, so these are its P largest): at least C+1 kept keys reach T1, hence every one of the best C+1 does;
    // (b) the keys >= T1 (a few more than C+1) are collected;  (c) and ranked among themselves by (value desc, slot asc).  More
    // collected keys than the list holds (runs of equal keys) -> nothing is proven for that query: state 2, exact fallback.
    // (On the 32-candidate form the two-level selection measured no faster — three more barriers and LDS atomics — and stays off.)
    if constexpr (C <= 32) {
        if (tid < QPW) sel_n[tid] = 0;
        const int qq = tid / TPQ, l16 = tid % TPQ;       // TPQ threads per query here
        for (int i = l16; i < SHARES * KEEP; i += TPQ) {
            const float vi = kept_v[qq][i];
            int rank = 0;
            for (int j = 0; j < SHARES * KEEP; ++j) {
                const float vj = kept_v[qq][j];
                rank += (vj > vi) || (vj == vi && j < i);
            }
            if (rank < C) { cand_key[qq][rank] = vi; cand_src[qq][rank] = kept_s[qq][i]; }
            if (rank == C) bound_rest[qq] = vi;
        }
        __syncthreads();
    } else {
        constexpr int P = (C + 1 + SHARES - 1) / SHARES;
        static_assert(P <= KEEP && SHARES * P >= C + 1, "the first P keys of every share must hold C+1 keys");
        const int qq = tid / TPQ, l16 = tid % TPQ;       // TPQ threads per query here
        for (int c = l16; c < C; c += TPQ) { cand_key[qq][c] = NEG; cand_src[qq][c] = -1; }
        if (l16 == 0) { bound_rest[qq] = NEG; sel_n[qq] = 0; thr1[qq] = NEG; }
        __syncthreads();
        for (int e = l16; e < SHARES * P; e += TPQ) {
            const float vi = kept_v[qq][(e / P) * KEEP + (e % P)];
            int rank = 0;
            for (int f = 0; f < SHARES * P; ++f) {
                const float vf = kept_v[qq][(f / P) * KEEP + (f % P)];
                rank += (vf > vi) || (vf == vi && f < e);
            }
            if (rank == C) thr1[qq] = vi;                     // ranks are a permutation: one writer
        }
        __syncthreads();
        const float t1 = thr1[qq];
        for (int i = l16; i < SHARES * KEEP; i += TPQ) {
            const float vi = kept_v[qq][i];
            if (vi >= t1 && vi > NEG) {
                const int pos = atomicAdd(&sel_n[qq], 1);
                if (pos < LMAX) { sel_v[qq][pos] = vi; sel_i[qq][pos] = i; }
            }
        }
        __syncthreads();
        const int ns = min(sel_n[qq], LMAX);
        for (int e = l16; e < ns; e += TPQ) {
            const float vi = sel_v[qq][e]; const int ii = sel_i[qq][e];
            int rank = 0;
            for (int f = 0; f < ns; ++f) {
                const float vf = sel_v[qq][f];
                rank += (vf > vi) || (vf == vi && sel_i[qq][f] < ii);
            }
            if (rank < C) { cand_key[qq][rank] = vi; cand_src[qq][rank] = kept_s[qq][ii]; }
            if (rank == C) bound_rest[qq] = vi;
        }
        __syncthreads();
    }

    // ---- 3. exact re-score of the candidates: thread (query, c) for c = share, share+16 ----
    const bool q_live = q < nq;
    const float* qv = queries + (size_t)(q_live ? q : 0) * dim;
    {
        float s2 = 0.f;
        for (int i = share; i < dim; i += SHARES) s2 += qv[i] * qv[i];
        qn2[ql][share] = s2;
    }
    for (int c = share; c < C; c += SHARES) {
        const int src = cand_src[ql][c];
        int row = -1; float d = __builtin_inff();
        if (src >= 0 && q_live && cand_key[ql][c] > NEG) {
            const uint32_t kb = __builtin_bit_cast(uint32_t, cand_key[ql][c]);
            const int64_t r = layout == 3 ? scan3_row_of(src >> 1, (int)(kb & 127u))
                            : layout == 2 ? scan2_row_of(src >> 1, (int)(kb & 127u)) : scan_row_of(src >> 1, (int)(kb & 127u));
            if (r < n_valid) { row = (int)r; d = 1.0f - exact_dot_chain_pf(rows + (size_t)r * dim, qv, dim); }
        }
        cand_row[ql][c] = row; cand_dist[ql][c] = d;
    }
    __syncthreads();

    // ---- 4. k-th best exact score so far; which bounds are still open? ----
    // 4a. the k-th smallest exact distance among the C candidates, selection by counting: thread (query, candidate) — one
    //     thread per query doing all C^2 comparisons held the workgroup for 215k of its 815k cycles
    {
        const int kk = k < C ? k : C;
        if (tid < QPW) dk_s[tid] = __builtin_inff();
        __syncthreads();
        for (int c = share; c < C; c += SHARES) {
            if (cand_row[ql][c] < 0) continue;
            const float di = cand_dist[ql][c]; const int ri = cand_row[ql][c];
            int rank = 0;
            for (int j = 0; j < C; ++j)
                rank += cand_row[ql][j] >= 0 && scored_before(tie, cand_dist[ql][j], cand_row[ql][j], di, ri);
            if (rank == kk - 1) dk_s[ql] = di;                          // rows are distinct: the ranks are a permutation, one writer
        }
        __syncthreads();
    }
    if (tid < QPW && q0 + tid < nq) {
        const int qq = tid;
        const int kk = k < C ? k : C;
        int have = 0;
        for (int i = 0; i < C; ++i) have += cand_row[qq][i] >= 0;
        const float dk = have >= kk ? dk_s[qq] : __builtin_inff();
        const float sk = 1.0f - dk;                            // k-th best exact score (−inf if fewer than k rows exist)
        float q2 = 0.f;
        for (int sh = 0; sh < SHARES; ++sh) q2 += qn2[qq][sh];
        const float SCAN_EPS = eps_rows * sqrtf(q2);           // NaN for a non-finite query: every test below fails -> exact fallback
        int st = 0;
        const bool all_rows_scored = n_valid <= 0;
        (void)all_rows_scored;
        if (have < kk || !(q2 >= SCAN_Q2_MIN && q2 <= SCAN_Q2_MAX) || sel_n[qq] > LMAX) {
            st = 2;                                            // fewer than k distinct rows among the candidates, or a query the fp16 bound does not cover
        } else {
            if (!(bound_rest[qq] + SCAN_EPS < sk)) st = 2;     // kept keys outside the best C could still matter
            for (int sh = 0; sh < SHARES; ++sh)
                if (!(share_floor[qq][sh] + SCAN_EPS < sk)) st = 2;   // a share dropped a key that could matter
            if (st == 0) {
                // a stream whose 2nd key is among the candidates hides rows bounded only by that key
                for (int c = 0; c < C; ++c) {
                    if ((cand_src[qq][c] & 1) && cand_key[qq][c] + SCAN_EPS >= sk) {
                        if (resc_n[qq] < RV_RESCAN_MAX) resc_stream[qq][resc_n[qq]++] = cand_src[qq][c] >> 1;
                        else st = 2;
                    }
                }
                if (st == 0 && resc_n[qq] > 0) st = 1;
            }
        }
        state[qq] = st;
        pool_n[qq] = C;
    }
    __syncthreads();

    // ---- 5. stream rescans: all 128 rows of each suspicious stream, exactly ----
    for (int qq = 0; qq < QPW; ++qq) {
        if (state[qq] != 1) continue;                           // block-uniform (LDS value)
        const int nres = resc_n[qq];
        const float* qv2 = queries + (size_t)(q0 + qq) * dim;
        // candidates of a re-scanned stream leave the pool (the stream's rows are all in it now): pool rows stay distinct
        for (int c = tid; c < C; c += 256)
            if (cand_row[qq][c] >= 0)
                for (int w = 0; w < nres; ++w) if ((cand_src[qq][c] >> 1) == resc_stream[qq][w]) cand_row[qq][c] = -1;
        for (int i = tid; i < nres * SCAN_STREAM_ROWS; i += 256) {
            const int which = i / SCAN_STREAM_ROWS, local = i - which * SCAN_STREAM_ROWS;
            const int64_t r = layout == 3 ? scan3_row_of(resc_stream[qq][which], local)
                            : layout == 2 ? scan2_row_of(resc_stream[qq][which], local) : scan_row_of(resc_stream[qq][which], local);
            int row = -1; float d = __builtin_inff();
            if (r < n_valid) { row = (int)r; d = 1.0f - exact_dot_chain_pf(rows + (size_t)r * dim, qv2, dim); }
            cand_row[qq][C + i] = row; cand_dist[qq][C + i] = d;
        }
        if (tid == 0) pool_n[qq] = C + nres * SCAN_STREAM_ROWS;
    }
    __syncthreads();

    // ---- 6. final exact top-k of the pool by (distance, row).  C <= 32: k rounds of "smallest key above the previous one"
    //         (a shuffle tree per round; duplicates would collapse).  Larger pools: rank counting — pool rows are distinct (step 5),
    //         so ranks are a permutation; the k rounds cost 21 us of a single query at k = 40 ----
    if constexpr (C <= 32) {
        const int qq = tid / TPQ, l16 = tid % TPQ;
        if (q0 + qq < nq) {
            const int pn = pool_n[qq];
            if (l16 == 0) flags[q0 + qq] = state[qq];
            uint64_t prev = 0; bool have_prev = false;
            for (int j = 0; j < k; ++j) {
                uint64_t best = ~0ull;
                for (int i = l16; i < pn; i += TPQ) {
                    if (cand_row[qq][i] < 0) continue;
                    const uint64_t key = dist_key(cand_dist[qq][i], tie_of(tie, cand_row[qq][i]));
                    if ((!have_prev || key > prev) && key < best) best = key;
                }
#pragma unroll
                for (int o = TPQ / 2; o > 0; o >>= 1) {
                    const uint64_t other = __shfl_xor(best, o, TPQ);
                    best = other < best ? other : best;
                }
                if (l16 == 0) {
                    const size_t o = (size_t)(q0 + qq) * k + j;
                    if (best == ~0ull) { out_ids[o] = -1; out_dist[o] = __builtin_inff(); }
                    else { out_ids[o] = tie_row(tie, (uint32_t)best); out_dist[o] = key_dist(best); }
                }
                prev = best; have_prev = true;
                if (best == ~0ull) {
                    for (int jj = j + 1 + l16; jj < k; jj += TPQ) {
                        out_ids[(size_t)(q0 + qq) * k + jj] = -1; out_dist[(size_t)(q0 + qq) * k + jj] = __builtin_inff();
                    }
                    break;
                }
            }
        }
    } else {
        const int qq = tid / TPQ, l16 = tid % TPQ;
        if (q0 + qq < nq) {
            const int pn = pool_n[qq];
            if (l16 == 0) flags[q0 + qq] = state[qq];
            int valid = 0;
            for (int i = 0; i < pn; ++i) valid += cand_row[qq][i] >= 0;                  // same count in every thread of the query
            for (int j = valid + l16; j < k; j += TPQ) { out_ids[(size_t)(q0 + qq) * k + j] = -1; out_dist[(size_t)(q0 + qq) * k + j] = __builtin_inff(); }
            for (int i = l16; i < pn; i += TPQ) {
                if (cand_row[qq][i] < 0) continue;
                const float di = cand_dist[qq][i]; const int ri = cand_row[qq][i];
                int rank = 0;
                for (int j = 0; j < pn; ++j)
                    rank += cand_row[qq][j] >= 0 && scored_before(tie, cand_dist[qq][j], cand_row[qq][j], di, ri);
                if (rank < k) { out_ids[(size_t)(q0 + qq) * k + rank] = cand_row[qq][i]; out_dist[(size_t)(q0 + qq) * k + rank] = cand_dist[qq][i]; }
            }
        }
    }
}

// the instantiation round 2 shipped, under its old name
// k <= 20: <8, 6, 32> (batches) / rescore_verify_small_kernel (<= 96 queries); k <= 64: <4, 10, 80>; k <= 100: <4, 12, 128> — the
// reference's API takes k up to 50 (src/api/routes.py:58) and searches the index for k * 2 (video_search_system.py:297)
constexpr int RV_K_SMALL = 20, RV_K_MID = 64, RV_K_MAX = 100;
constexpr int RVL_QPW = 4, RVL_KEEP = 10, RVL_C = 80;
constexpr int RVX_QPW = 4, RVX_KEEP = 12, RVX_C = 128;
static const auto rescore_verify_kernel = rescore_verify_kernel_t<RV_QPW, RV_KEEP, RV_C>;
static const auto rescore_verify_large_kernel = rescore_verify_kernel_t<RVL_QPW, RVL_KEEP, RVL_C>;
static const auto rescore_verify_xlarge_kernel = rescore_verify_kernel_t<RVX_QPW, RVX_KEEP, RVX_C>;
// small batches (the streaming scan's query-major keys) with k > RV_K_SMALL: ONE query per workgroup, so that 256 threads instead
// of 64 walk the query's 2 x streams keys (one query, k = 40, 1M rows: 0.36 ms with the four-queries-per-workgroup form above)
static const auto rescore_verify_large1_kernel = rescore_verify_kernel_t<1, RVL_KEEP, RVL_C>;
static const auto rescore_verify_xlarge1_kernel = rescore_verify_kernel_t<1, RVX_KEEP, RVX_C>;

// ---------------------------------------------------------------------------------------------------------------
// Small batches (nq <= SCAN3_MAX_Q, keys in layout 3): ONE workgroup per query instead of 16 queries per workgroup.
// rescore_verify_kernel walks streams/16 keys per thread (488 dependent steps at 1M rows: 0.26 ms for a single
// query, more than the HBM-bound scan in front of it); here the 256 threads of a workgroup split one query's
// streams 256 ways (31 steps at 1M rows).  Same proof, same outcome codes, same exact arithmetic:
//   1. pass A over the keys: each thread's largest key; T = min over the waves of the wave's 8th largest thread maximum
//      (>= 32 keys reach T, so the best RV_C do); pass B: the keys >= T are collected (a few dozen), the largest key
//      below T bounds everything else
//   2. the collected keys are ranked; the best RV_C become candidates, the next one joins the bound
//   3. exact fp64-chain re-score of the candidates from the fp32 master (one thread per candidate), the best
//      max(16, k + 6) first, the others only if their best key could still reach the k-th exact score
//   4. k-th best exact score s_k; every bound + eps must lie below it; a stream whose 2nd key could still matter
//      is re-scored in full (<= RV_RESCAN_MAX), its candidates dropped from the pool (no duplicates)
//   5. exact top-k of the pool by (distance, row): rank counting

#ifdef VQ_GEMM_TOWER_STAMPS      // `make STAMPS=1` diagnostic build only (scripts/rescore_stamps.py): phase boundaries of query 0's workgroup
__device__ unsigned long long g_rs_stamps[16];
#define VQ_RS_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_rs_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VQ_RS_STAMP(i)
#endif

template <int C>
__global__ __launch_bounds__(256)
void rescore_verify_small_kernel_t(const uint32_t* __restrict__ keys, int64_t streams, int64_t q_pad,
                                 const float* __restrict__ rows, int64_t n_valid, int dim,
                                 const float* __restrict__ queries, int nq, int k,
                                 int32_t* __restrict__ out_ids, float* __restrict__ out_dist,
                                 int32_t* __restrict__ flags, float eps_rows,
                                 int32_t* __restrict__ slots1 /* with counters1: a ONE-query launch writes the flagged list and */,
                                 int32_t* __restrict__ counters1 /* the outcome counters itself (collect_flags_kernel's job) */,
                                 const TieOrder tie /* (distance, id) order of the result: vq_common.h */) {
    constexpr int POOL = C + RV_RESCAN_MAX * SCAN_STREAM_ROWS;      // exact-scored rows per query
    constexpr int TPR = 256 / C;                                   // threads staging one candidate row
    static_assert(C == 32 || C == 64, "candidates re-scored exactly: 32 (k <= 20) or 64 (k <= 40)");
    __shared__ __attribute__((aligned(8))) int2 sel[4 * C];      // {key bits, source}: one 8-byte LDS access per entry
    __shared__ float wave_floor[4];
    __shared__ float red[256];
    __shared__ float cand_key[C];
    __shared__ int cand_src[C];
    __shared__ int cand_row[POOL];
    __shared__ float cand_dist[POOL];
    __shared__ float bound_rest_s, qnorm2_s, dk_s;
    __shared__ int resc_stream[RV_RESCAN_MAX];
    __shared__ int resc_n, state, pool_n, have_s, surv_n;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    if (q >= nq) return;
    const float NEG = -__builtin_inff();
    const float* qv = queries + (size_t)q * dim;

    VQ_RS_STAMP(0);
    // ---- 1. the best C of the query's 2*streams keys, and a bound on all the others.  The keys (query-major: one contiguous
    //         run, L2-resident) are read TWICE instead of each thread keeping a sorted short list of its share: pass A takes each
    //         thread's largest key; T = the smallest, over the four waves, of the wave's 8th largest thread maximum, so at least
    //         32 keys are >= T and the overall best C are among them; pass B collects the keys >= T (a few dozen) and the
    //         largest key below T.  (Stamps: the insertion lists cost ~30 instructions per key whenever ANY lane of the wave
    //         inserted — 23k cycles; ranking the 256 maxima against each other another ~10k.) ----
    constexpr int PF = 16, KEEP_ROUNDS = 2;
    const uint2 NONE = uint2{0xFF800000u, 0xFF800000u};      // a missing stream reads as -inf
    const uint32_t* qkeys = keys + (size_t)q * streams * 2;    // layout 3: query-major
    // thread -> first stream: by thread id (a wave reads 512 contiguous bytes); on a small index interleaved over the waves,
    // so that every wave owns at least 8 streams as soon as the keys no longer all fit the collected list
    const int p0 = streams < 512 ? 4 * lane + wave : tid;
    // up to KEEP_ROUNDS x PF x 256 streams (1M rows) the keys of pass A STAY in registers for pass B: the second read and its
    // two round trips (stamps: pass B was 26k of the kernel's 73k cycles) are gone
    const bool keep = streams <= (int64_t)KEEP_ROUNDS * PF * 256;      // workgroup-uniform
    uint2 kept[KEEP_ROUNDS][PF];
    auto load_round = [&](int64_t s, uint2 (&two)[PF]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PF; ++u) { const int64_t su = s + u * 256; two[u] = su < streams ? *(const uint2*)(qkeys + su * 2) : NONE; }
    };
    float mx = NEG;
    if (keep) {
#pragma unroll
        for (int r = 0; r < KEEP_ROUNDS; ++r) {
            load_round(p0 + (int64_t)r * PF * 256, kept[r]);
#pragma unroll
            for (int u = 0; u < PF; ++u) mx = fmaxf(mx, __builtin_bit_cast(float, kept[r][u].x));      // a stream's 2nd key is never above its 1st
        }
    } else {
        for (int64_t s = p0; s < streams; s += PF * 256) {
            uint2 two[PF];
            load_round(s, two);
#pragma unroll
            for (int u = 0; u < PF; ++u) mx = fmaxf(mx, __builtin_bit_cast(float, two[u].x));
        }
    }
    VQ_RS_STAMP(1);
    {   // the wave's 8th largest maximum: take the wave maximum eight times, retiring one holder each time.  (Touching the retiring
        // lanes' rows of the fp32 master here, by LDS-DMA into a scratch word, so that step 3's 16 random 2-KiB rows would find
        // their translations ready, did not move step 3's 19k cycles: removed again.)
        float cur = mx, t8 = NEG;
#pragma unroll
        for (int r = 0; r < C / 4; ++r) {                  // the wave's (C / 4)-th largest: 4 waves x C / 4 = C keys reach the threshold
            t8 = wave64_max(cur);
            const unsigned long long holders = __ballot(cur == t8);
            if (holders && lane == (int)__builtin_ctzll(holders)) cur = NEG;      // (all -inf: lane 0 "retires", nothing changes; none only if every key is NaN)
        }
        if (lane == 0) wave_floor[wave] = t8;
    }
    {   // |q|^2: wave sums, then four partials through `red`
        float s2 = 0.f;
        for (int i = tid; i < dim; i += 256) s2 += qv[i] * qv[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
        if (lane == 0) red[wave] = s2;
    }
    if (tid < C) { cand_key[tid] = NEG; cand_src[tid] = -1; }
    if (tid == 0) { resc_n = 0; state = 0; surv_n = 0; bound_rest_s = NEG; }
    __syncthreads();
    VQ_RS_STAMP(2);
    const float thr = fminf(fminf(wave_floor[0], wave_floor[1]), fminf(wave_floor[2], wave_floor[3]));
    if (tid == 0) qnorm2_s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();                                           // wave_floor is rewritten below
    {
        float below = NEG;                                     // best key NOT collected
        const float thr_eff = fmaxf(thr, -3.4e38f);            // thr = -inf on a tiny index: every real key (they are finite), never a missing one
        // one round of 32 keys per lane: which of them the lane collects as a bit mask, the lanes' counts scanned over the wave
        // (six shuffles), ONE LDS atomic per wave and round for the wave's run in the list, then the predicated stores.  (A ballot, an
        // atomic and a shuffle per key slot were 28k cycles of this kernel; a scalar walk over the collecting lanes ~1k per round.)
        auto collect_round = [&](int64_t s, const uint2 (&two)[PF]) __attribute__((always_inline)) {
            unsigned km = 0;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const float v0 = __builtin_bit_cast(float, two[u].x), v1 = __builtin_bit_cast(float, two[u].y);
                const bool k0 = v0 >= thr_eff, k1 = v1 >= thr_eff;
                km |= (k0 ? 1u : 0u) << (2 * u) | (k1 ? 1u : 0u) << (2 * u + 1);
                below = fmaxf(below, fmaxf(k0 ? NEG : v0, k1 ? NEG : v1));
            }
            if (__ballot(km != 0) == 0) return;                // wave-uniform
            const int mine = __builtin_popcount(km);
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
            const int total = __builtin_amdgcn_readlane(incl, 63);
            int base = 0;
            if (lane == 0) base = atomicAdd(&surv_n, total);
            int pos = __builtin_amdgcn_readfirstlane(base) + incl - mine;
            if (km) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        if (km >> (2 * u + w) & 1u) {
                            if (pos < 4 * C) sel[pos] = int2{(int)(w ? two[u].y : two[u].x), (int)((s + u * 256) * 2) + w};
                            ++pos;
                        }
                    }
                }
            }
        };
        if (keep) {
#pragma unroll
            for (int r = 0; r < KEEP_ROUNDS; ++r) collect_round(p0 + (int64_t)r * PF * 256, kept[r]);
        } else {
            for (int64_t s = p0; s < streams; s += PF * 256) {
                uint2 two[PF];
                load_round(s, two);
                collect_round(s, two);
            }
        }
        below = wave64_max(below);
        if (lane == 0) wave_floor[wave] = below;
    }
    __syncthreads();
    {
        const int ns = min(surv_n, 4 * C);                                // more keys than the list holds reach the threshold: flagged in step 4
        if (tid < ns) {
            const float vi = __builtin_bit_cast(float, sel[tid].x); const int si = sel[tid].y;
            int rank = 0;
            // the list is read as a broadcast, eight entries per trip in flight (one dependent LDS round trip per entry was 6-13k cycles)
            int j2 = 0;
            for (; j2 + 8 <= ns; j2 += 8) {
                int2 e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = sel[j2 + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float vj = __builtin_bit_cast(float, e[u].x);
                    rank += (vj > vi) || (vj == vi && e[u].y < si);            // sources are distinct: ranks are a permutation
                }
            }
            for (; j2 < ns; ++j2) {
                const int2 e = sel[j2];
                const float vj = __builtin_bit_cast(float, e.x);
                rank += (vj > vi) || (vj == vi && e.y < si);
            }
            if (rank < C) { cand_key[rank] = vi; cand_src[rank] = si; }
            if (rank == C) bound_rest_s = vi;
        }
    }
    __syncthreads();
    if (tid == 0)       // what was not selected: the (C+1)-th collected key and every key below the threshold
        bound_rest_s = fmaxf(fmaxf(bound_rest_s, fmaxf(wave_floor[0], wave_floor[1])), fmaxf(wave_floor[2], wave_floor[3]));
    __syncthreads();

    VQ_RS_STAMP(3);
    // ---- 3. exact re-score of the candidates: the rows (and the query) are staged into LDS by all 256 threads
    //         in one round of wide loads — a thread walking its own 2-KiB row straight from HBM pays one exposed
    //         round trip per 16 bytes — then one thread per candidate runs the fixed-order fp64 chain out of LDS.
    //         Two passes: a random row of the fp32 master costs an address translation (~1.1k cycles each, served one after the
    //         other: 32 rows = 35k cycles, stamps), so the best `c1` candidates go first and the others follow only when the
    //         best key among them could still reach the k-th exact score found so far. ----
    if (tid < C) {
        const int src = cand_src[tid];
        int row = -1;
        if (src >= 0 && cand_key[tid] > NEG) {
            const uint32_t kb = __builtin_bit_cast(uint32_t, cand_key[tid]);
            const int64_t r = scan3_row_of(src >> 1, (int)(kb & 127u));
            if (r < n_valid) row = (int)r;
        }
        cand_row[tid] = row;
        cand_dist[tid] = __builtin_inff();
    }
    const int kk = k < C ? k : C;
    const int c1 = min(C, max(16, (k + 6 + 7) & ~7));        // first pass: k candidates and a margin, whole groups of 8
    int limit = c1, my_rank = -1;                                // candidates re-scored; this thread's candidate among them by (distance, row)
    extern __shared__ __attribute__((aligned(16))) float rs_dyn[];        // [C][dim + 4] rows, then [dim] query
    const int ldr = dim + 4, d4 = dim >> 2;                                // +4 floats: rows on different LDS banks
    float* qbuf = rs_dyn + C * ldr;
    static_assert(C * TPR == 256, "TPR threads stage one candidate row");
    for (int pass = 0; pass < 2; ++pass) {
        const int lo = pass == 0 ? 0 : c1, hi = pass == 0 ? c1 : C;
        if (tid == 0) dk_s = __builtin_inff();
        __syncthreads();                                         // cand_row / the decision of the previous pass visible
        {   // thread = (candidate c, eighth e): float4 e, e + 8, ... of the row, 16 loads issued before the first LDS store
            const int c = tid / TPR, e = tid % TPR;
            const int r = c >= lo && c < hi ? cand_row[c] : -1;
            if (r >= 0) {
                const float4* src = (const float4*)(rows + (size_t)r * dim);
                float4* dst = (float4*)(rs_dyn + c * ldr);
                for (int j0 = e; j0 < d4; j0 += TPR * 16) {
                    float4 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = src[min(j0 + TPR * u, d4 - 1)];      // always loaded (clamped): a conditionally
#pragma unroll                                                                              // written array goes to scratch
                    for (int u = 0; u < 16; ++u) if (j0 + TPR * u < d4) dst[j0 + TPR * u] = v[u];
                }
            }
        }
        if (pass == 0) {
            VQ_RS_STAMP(4);
            for (int i = tid; i < d4; i += 256) *(float4*)(qbuf + 4 * i) = *(const float4*)(qv + 4 * i);
        }
        __syncthreads();
        if (pass == 0) VQ_RS_STAMP(5);
        if (tid >= lo && tid < hi && cand_row[tid] >= 0) cand_dist[tid] = 1.0f - exact_dot_chain(rs_dyn + tid * ldr, qbuf, dim);
        __syncthreads();

        // ---- 4. k-th best exact distance among the candidates re-scored so far, by rank counting (all in wave 0) ----
        my_rank = -1;
        if (tid < hi && cand_row[tid] >= 0) {
            const float di = cand_dist[tid]; const int ri = cand_row[tid];
            int rank = 0;
#pragma unroll 8
            for (int j = 0; j < hi; ++j)
                rank += cand_row[j] >= 0 && scored_before(tie, cand_dist[j], cand_row[j], di, ri);
            if (rank == kk - 1) dk_s = di;
            my_rank = rank;
        }
        if (wave == 0) {
            const int have = __builtin_popcountll(__ballot(my_rank >= 0));
            if (lane == 0) have_s = have;
        }
        __syncthreads();
        limit = hi;
        if (hi == C) break;
        {   // the best key not re-scored yet bounds every candidate behind it (cand_key is sorted): can it still matter?
            const float dk = have_s >= kk ? dk_s : __builtin_inff();
            const float nextkey = cand_key[hi];
            const bool more = nextkey > NEG && !(nextkey + eps_rows * sqrtf(qnorm2_s) < 1.0f - dk);     // same values in every thread
            __syncthreads();                                     // ... and every thread has read them before pass 1 resets dk_s
            if (!more) break;
        }
    }
    VQ_RS_STAMP(6);
    if (wave == 0) {       // the verdict: every lane tests its own candidate, lane 0 collects (was one thread walking all C)
        const int have = have_s;
        const float dk = have >= kk ? dk_s : __builtin_inff();
        const float sk = 1.0f - dk;
        const float eps = eps_rows * sqrtf(qnorm2_s);          // NaN for a non-finite query: every test fails -> exact fallback
        // a stream's 2nd key: its unseen rows are bounded only by it (candidates not re-scored lie below sk - eps: never set)
        const bool second = tid < limit && (cand_src[tid] & 1) && cand_key[tid] + eps >= sk;
        unsigned long long need = __ballot(second);
        if (tid >= limit && tid < C) cand_row[tid] = -1;    // not re-scored: not part of the pool
        if (lane == 0) {
            int st = 0, nres = 0;
            if (have < kk || surv_n > 4 * C || !(qnorm2_s >= SCAN_Q2_MIN && qnorm2_s <= SCAN_Q2_MAX)) st = 2;      // (more keys tied at the selection threshold than the list holds; a query the fp16 bound does not cover)
            else {
                if (!(bound_rest_s + eps < sk)) st = 2;            // a key outside the best C could still matter
                if (st == 0) {
                    while (need) {                                   // ascending candidate order, as the serial walk took them
                        const int c = __builtin_ctzll(need);
                        need &= need - 1;
                        if (nres < RV_RESCAN_MAX) resc_stream[nres++] = cand_src[c] >> 1; else st = 2;
                    }
                    if (st == 0 && nres > 0) st = 1;
                }
            }
            state = st; resc_n = st == 1 ? nres : 0;
        }
    }
    __syncthreads();

    VQ_RS_STAMP(7);
    // ---- 5. stream rescans (all 128 rows, exactly); candidates of a rescanned stream leave the pool ----
    const int nres = resc_n;
    if (nres > 0) {
        if (tid < C && cand_row[tid] >= 0) {
            for (int w = 0; w < nres; ++w) if ((cand_src[tid] >> 1) == resc_stream[w]) cand_row[tid] = -1;
        }
        for (int i = tid; i < nres * SCAN_STREAM_ROWS; i += 256) {
            const int which = i / SCAN_STREAM_ROWS, local = i - which * SCAN_STREAM_ROWS;
            const int64_t r = scan3_row_of(resc_stream[which], local);
            int row = -1; float d = __builtin_inff();
            if (r < n_valid) { row = (int)r; d = 1.0f - ((dim & 31) == 0 ? exact_dot_chain_pf(rows + (size_t)r * dim, qv, dim) : exact_dot_chain(rows + (size_t)r * dim, qv, dim)); }   // _pf: eight 16-byte loads in flight per thread (a stream rescan was ~35 us of one-load-per-step round trips)
            cand_row[C + i] = row; cand_dist[C + i] = d;
        }
    }
    if (tid == 0) pool_n = C + nres * SCAN_STREAM_ROWS;
    __syncthreads();

    VQ_RS_STAMP(8);
    // ---- 6. exact top-k of the pool: rank by (distance, row); rows are distinct, so ranks are a permutation ----
    if (nres == 0) {       // no rescans: the pool is the candidate list and step 4's ranks are the answer
        const int have = have_s;
        if (tid == 0) flags[q] = state;
        for (int j = tid; j < k; j += 256)
            if (j >= have) { out_ids[(size_t)q * k + j] = -1; out_dist[(size_t)q * k + j] = __builtin_inff(); }
        if (my_rank >= 0 && my_rank < k) { out_ids[(size_t)q * k + my_rank] = cand_row[tid]; out_dist[(size_t)q * k + my_rank] = cand_dist[tid]; }
    } else {
        const int pn = pool_n;
        if (tid == 0) flags[q] = state;
        for (int j = tid; j < k; j += 256) { out_ids[(size_t)q * k + j] = -1; out_dist[(size_t)q * k + j] = __builtin_inff(); }
        __syncthreads();
        for (int i = tid; i < pn; i += 256) {
            if (cand_row[i] < 0) continue;
            const float di = cand_dist[i]; const int ri = cand_row[i];
            int rank = 0;
            for (int j = 0; j < pn; ++j)
                rank += cand_row[j] >= 0 && scored_before(tie, cand_dist[j], cand_row[j], di, ri);
            if (rank < k) { out_ids[(size_t)q * k + rank] = cand_row[i]; out_dist[(size_t)q * k + rank] = cand_dist[i]; }
        }
    }
    if (counters1 && tid == 0) {        // nq == 1: one workgroup saw the only flag — one launch fewer on the single-query path
        const int st = state;
        if (st == 2) slots1[0] = 0;
        counters1[0] = st == 2; counters1[1] = st == 0; counters1[2] = st == 1; counters1[3] = st == 2;
    }
    VQ_RS_STAMP(9);
}

// k <= RV_K_SMALL: 32 candidates (what round 2 shipped, under its old name); RV_K_SMALL < k <= RV_K_SMALL64 on a one-query-at-a-time
// search (the caller's k * 2 for a user k of 11 .. 20, video_search_system.py:297): 64 candidates — a lone workgroup has the CU's LDS
// to itself (64 staged rows of 512 floats = 132 KB), where the wide-pool kernel above spends 0.06 ms on one query
constexpr int RV_K_SMALL64 = 40;
static const auto rescore_verify_small_kernel = rescore_verify_small_kernel_t<32>;
static const auto rescore_verify_small64_kernel = rescore_verify_small_kernel_t<64>;

}  // namespace vq

// Persistent variant of the 256x256 four-phase GEMM (gemm_mfma256.h): one workgroup per CU walks a
// list of output tiles with ONE continuous, software-pipelined K loop, so that
//   * the first K-tiles of tile i+1 are already in flight while tile i finishes (no per-tile prologue
//     round trip: with K = 768 a tile is only 12 K-tiles long and the fixed per-tile cost was ~2/3 of
//     its time in the one-tile-per-workgroup kernel), and
//   * tile i's epilogue stores are issued from inside tile i+1's first phases and drain under its MFMAs.
//
// Same geometry, LDS image, phases, stagger and hazards as gemm_tn256_kernel; additions:
//   - LDS = 128 KiB ring + 8 x 4 KiB epilogue strips (160 KiB total).  A strip holds 16 rows x 64 cols
//     fp32 with the 16-byte chunk XOR-swizzled by the row (chunk ^ row), conflict-free for the transposing
//     b128 write and the row-wise b128 read.
//   - a finished m-half of the wave tile (64 rows x 64 cols, both n quadrants done after phase 2 resp.
//     phase 4 of a tile's last K-tile) is written out in the READ half of the following phase (phase 3 of
//     that K-tile, resp. phase 1 of the next tile's first K-tile), i.e. under the partner group's MFMAs.
//   - epilogue stores/loads count in vmcnt and are OLDER than the two DMA pieces issued after them in the
//     same read half, so the counted vmcnt(2) of phase 4 still means "everything but the newest half-tile
//     has landed" (it additionally waits for the stores' acknowledgements once per tile).
//   - tile order: an XCD's 32 concurrent workgroups take a 4 (m) x 8 (n) block of tiles per round when
//     tiles_n >= 8 (else (32/tiles_n) x tiles_n), so operand panels are shared in that XCD's L2.
#pragma once
#include "vq_common.h"
#include <cstring>
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "gemm_mfma160.h"
#include "gemm_mfma256w4.h"
#include "gemm_mfma256d.h"
#include "gemm_mfma128x256.h"
#ifdef VQ_GEMM_EXPERIMENTS
#include "gemm_mfma128x256p.h"
#endif
#include "gemm_mfma256f.h"
#ifdef VQ_DIAG       // the hand-scheduled four-wave kernel (id 24) changed nothing in frames/s (DESIGN.md section 4 "Round 3" (5)): diagnostic builds only
#include "gemm_asm256.h"
#endif

namespace vq {

constexpr int G2P_STRIP = 16 * 256;                         // 4 KiB per wave
constexpr int G2P_LDS_BYTES = G2_LDS_BYTES + 8 * G2P_STRIP; // 160 KiB

static inline int blocked_tile_slots(int tiles_m, int tiles_n) { return tiles_m * tiles_n; }

template <bool IS_F16, class Epi>
__global__ __launch_bounds__(G2_THREADS, 2)
void gemm_tn256p_kernel(const uint16_t* __restrict__ A, int lda,
                        const uint16_t* __restrict__ W, int ldw,
                        int K, int tiles_m, int tiles_n, int slots, Epi epi) {
    typedef mfma_op<IS_F16> op;
    typedef typename op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // this workgroup's tile list: round r -> logical slot (r*8 + xcd)*32 + slot_in_xcd   (blocks b and b+8 share an XCD)
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, in_xcd = blockIdx.x >> 3;
    const int per_xcd = G >> 3;                                    // G is a multiple of 8
    auto slot_of = [&](int r) { return (r * 8 + xcd) * per_xcd + in_xcd; };
    // count the valid tiles first (wave-uniform), so the flattened loop length is known
    int my_tiles = 0;
    for (int r = 0; slot_of(r) < slots; ++r) {
        int tm, tn;
        if (tile_coords(slot_of(r), tiles_m, tiles_n, tm, tn)) ++my_tiles;
    }
    if (my_tiles == 0) return;                                     // whole workgroup leaves before any barrier

    const int srow = lane >> 3, sslot = lane & 7;
    int a_rowoff[2], w_rowoff[2];                                  // element offsets inside a half-tile panel
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + srow;
        const int chunk = sslot ^ ((row >> 1) & 7);
        a_rowoff[i] = row * lda + chunk * 8;
        w_rowoff[i] = row * ldw + chunk * 8;
    }
    const size_t a_half = (size_t)128 * lda, w_half = (size_t)128 * ldw;
    const int piece_off = wave * 2048;
    const int nk = K / G2_BK;
    const int total = my_tiles * nk;                               // flattened K-tiles

    // prefetch cursor (runs ahead of the compute cursor, crosses tile boundaries first)
    int pf_round = -1, pf_k = nk;                                  // forces a tile fetch on first use
    const uint16_t* pf_a = A;
    const uint16_t* pf_w = W;
    auto pf_advance_tile = [&]() __attribute__((always_inline)) {
        int tm = 0, tn = 0;
        do { ++pf_round; } while (!tile_coords(slot_of(pf_round), tiles_m, tiles_n, tm, tn));
        pf_a = A + (size_t)tm * G2_BM * lda;
        pf_w = W + (size_t)tn * G2_BN * ldw;
        pf_k = 0;
    };
    // stage one half-tile of the K-tile the prefetch cursor points at; `advance` moves the cursor afterwards
    auto stage = [&](int buf, int which) __attribute__((always_inline)) {
        char* dst = smem + buf * G2_BUF + which * G2_HALF + piece_off;
        const int koff = pf_k * G2_BK;
        if (which < 2) {
            const uint16_t* base = pf_a + (which ? a_half : 0) + koff;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + a_rowoff[0]), (lds_void_t*)(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + a_rowoff[1]), (lds_void_t*)(dst + 1024), 16, 0, 0);
        } else {
            const uint16_t* base = pf_w + ((which & 1) ? w_half : 0) + koff;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + w_rowoff[0]), (lds_void_t*)(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(base + w_rowoff[1]), (lds_void_t*)(dst + 1024), 16, 0, 0);
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
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // write out m-half hm (64 rows x 64 cols) of the wave tile of output tile (m0, n0) and clear it
    char* strip = smem + G2_LDS_BYTES + wave * G2P_STRIP;
    const int rrow = lane >> 4, rcol = lane & 15;
    auto epilogue_half = [&](int hm, int m0, int n0) __attribute__((always_inline)) {
        const int ncol = n0 + wc * 64 + rcol * 4;
        const f32x4 bias = epi.bias_at(ncol);
#pragma unroll
        for (int i = 0; i < 4; ++i) {                 // one 16-row m tile per pass
            const int mi = hm * 4 + i;
            f32x4 loaded[4];
            if constexpr (Epi::kLoads) {
#pragma unroll
                for (int it = 0; it < 4; ++it) loaded[it] = epi.load(m0 + wr * 128 + mi * 16 + it * 4 + rrow, ncol);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                *(f32x4*)(strip + frow * 256 + (((ni * 4 + fgrp) ^ frow) & 15) * 16) = acc[mi][ni];
                acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + rrow;
                const f32x4 v = *(const f32x4*)(strip + row * 256 + ((rcol ^ row) & 15) * 16);
                epi.store(m0 + wr * 128 + mi * 16 + row, ncol, v, bias, Epi::kLoads ? loaded[it] : f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
    };

    // compute cursor
    int c_round = -1, c_k = 0, c_m0 = 0, c_n0 = 0, prev_m0 = 0, prev_n0 = 0;
    auto c_advance_tile = [&]() __attribute__((always_inline)) {
        int tm = 0, tn = 0;
        do { ++c_round; } while (!tile_coords(slot_of(c_round), tiles_m, tiles_n, tm, tn));
        prev_m0 = c_m0; prev_n0 = c_n0;
        c_m0 = tm * G2_BM; c_n0 = tn * G2_BN;
    };

    // one K-tile: same four phases as gemm_tn256_kernel.  first = first K-tile of a tile (write out the
    // previous tile's m-half 1 in phase 1), last = last K-tile (write out m-half 0 in phase 3).
    // pf_left = flattened K-tiles not yet staged.
    int pf_left = total;
    auto stage_next = [&](int buf, int which) __attribute__((always_inline)) {
        // stages half-tile `which` of the K-tile under the prefetch cursor; the cursor moves after which == 3
        if (which == 0 && pf_k == nk) pf_advance_tile();
        stage(buf, which);
        if (which == 3) { ++pf_k; --pf_left; }
    };
    auto tile = [&](int bufi, bool first, bool last, bool have_prev) __attribute__((always_inline)) {
        const char* buf = smem + bufi * G2_BUF;
        // Prefetch bookkeeping: when this K-tile starts, the NEXT K-tile's A half 0 is already staged
        // (which = 0 was issued in phase 4 of the previous K-tile), so phases 1-3 stage which = 1,2,3 of it.
        const bool next = pf_left > 0;                 // a K-tile after this one exists
        if (first && have_prev) epilogue_half(1, prev_m0, prev_n0);
        load_a(buf, 0); load_w(buf, 0);
        if (next) stage_next(bufi ^ 1, 1);
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 0);
        barrier();
        load_w(buf, 1);
        if (next) stage_next(bufi ^ 1, 2);
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(0, 1);
        barrier();
        if (last) epilogue_half(0, c_m0, c_n0);
        load_a(buf, 1);
        if (next) stage_next(bufi ^ 1, 3);             // cursor now points at the K-tile after next
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 1);
        barrier();
        load_w(buf, 0);
        if (pf_left > 0) { stage_next(bufi, 0); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        else             { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mfma_quadrant(1, 0);
        barrier();
    };

    // ---- prologue: K-tile 0 complete, K-tile 1's A half 0 in flight ----
    stage_next(0, 0); stage_next(0, 1); stage_next(0, 2); stage_next(0, 3);
    if (pf_left > 0) { stage_next(1, 0); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else             { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    barrier();

    if (wr == 1) barrier();
    bool have_prev = false;
    for (int kt = 0; kt < total; kt += 2) {              // nk is even: tiles start on even kt
        const bool first = c_k == 0;
        if (first) c_advance_tile();
        tile(0, first, false, have_prev);
        ++c_k;
        const bool last = c_k + 1 == nk;
        tile(1, false, last, false);
        if (++c_k == nk) { c_k = 0; have_prev = true; }
    }
    if (wr == 0) barrier();
    epilogue_half(1, c_m0, c_n0);                         // the last tile's second m-half
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn256p(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                              int M, int N, int K, const Epi& epi, int num_cus = 256) {
    VQ_CHECK(M > 0 && M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0,
             "gemm_tn256p: shape M=%d N=%d K=%d is not tile-aligned (256/256/128)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn256p: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    VQ_CHECK((int64_t)128 * lda < ((int64_t)1 << 31) && (int64_t)128 * ldw < ((int64_t)1 << 31), "gemm_tn256p: panel too wide");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256p_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, G2P_LDS_BYTES));
        attr_set = true;
    }
    const int tiles_m = M / G2_BM, tiles_n = N / G2_BN;
    const int slots = blocked_tile_slots(tiles_m, tiles_n);
    int grid = std::min(num_cus, (int)round_up(slots, 8));
    grid = grid / 8 * 8;
    hipLaunchKernelGGL((gemm_tn256p_kernel<IS_F16, Epi>), dim3(grid), dim3(G2_THREADS), G2P_LDS_BYTES, st,
                       A, lda, W, ldw, K, tiles_m, tiles_n, slots, epi);
    VQ_HIP(hipGetLastError());
    return 0;
}

// Dispatch: the phased 256x256 kernel when the problem tiles by it and yields enough
// workgroups to occupy the chip, else the 128x128 kernel.  force: 1 = 128x128, 2 = four-phase, 8 = four-phase with the deep prefetch and buffer_load..lds staging (gemm_mfma256d.h, the default 256x256 mainloop), 11 = the same with global_load_lds staging, 3 = ring,
// 4 = persistent four-phase, 5 = 160x256 ring, 6 = auto without the 160-row tiles, 7 = four-wave 256x256.  Auto picks the 160-row tiles when they put one workgroup on
// more CUs than 256-row tiles would (VQ_AMD_GEMM160=0 disables that).
static inline int gemm_use_deep() {           // $VQ_AMD_GEMM256=4phase: auto picks the second-generation mainloop (A/B switch)
    static int v = -1;
    if (v < 0) { const char* e = getenv("VQ_AMD_GEMM256"); v = (e && !strcmp(e, "4phase")) ? 0 : 1; }
    return v;
}
template <bool IS_F16, class Epi>
static int launch_gemm_tn256_best(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                                  int M, int N, int K, const Epi& epi) {
    return gemm_use_deep() ? launch_gemm_tn256d<IS_F16>(st, A, lda, W, ldw, M, N, K, epi)
                           : launch_gemm_tn256<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
}

static inline int gemm_multi_min_wgs() {        // $VQ_AMD_GEMM_MULTI_MIN: fewest workgroups a three-tile launch may leave.  Default 128 [r03]: with
    static int v = -1;                           // three batches in flight qkv (450 tiles -> 150 workgroups) gains 0.6-0.9 % frames/s too (192 kept it on single tiles)
    if (v < 0) { const char* e = getenv("VQ_AMD_GEMM_MULTI_MIN"); v = e ? atoi(e) : 128; }
    return v;
}

static inline bool gemm_use_multi() {          // $VQ_AMD_GEMM_MULTI=0: one tile per workgroup everywhere
    static int v = -1;
    if (v < 0) { const char* e = getenv("VQ_AMD_GEMM_MULTI"); v = (e && atoi(e) == 0) ? 0 : 1; }
    return v != 0;
}

static inline bool gemm_use_tail_split() {     // $VQ_AMD_GEMM_TAIL=0 keeps one launch per GEMM
    static int v = -1;
    if (v < 0) { const char* e = getenv("VQ_AMD_GEMM_TAIL"); v = (e && atoi(e) == 0) ? 0 : 1; }
    return v != 0;
}

template <bool IS_F16, class Epi>
static int launch_gemm_auto(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                            int M, int N, int K, const Epi& epi, int force = 0) {
    // 24: the hand-scheduled four-wave 256x256 mainloop (gemm_asm256.h) on every shape that tiles
    if (force == 24) {
#ifdef VQ_DIAG
        if (M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0) return launch_gemm_tn256a<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
        force = 6;
#else
        return fail(VQ_ERR_INVALID, "gemm kernel 24 (hand-scheduled four-wave loop) is built into diagnostic libraries only: `make DIAG=1 OUT=... OBJDIR=...`");
#endif
    }
    // 20 / 21: persistent out-of-phase 128x256 tiles, two workgroups per CU (gemm_mfma128x256p.h), on every shape that tiles
    // (20: the second workgroup of a CU starts half a tile late; 21: no lag — the in-step control of the A/B)
#ifdef VQ_GEMM_EXPERIMENTS
    if ((force == 20 || force == 21) && M % GP_BM == 0 && N % GP_BN == 0 && K % (2 * GP_SUB_K) == 0 && K >= 4 * GP_SUB_K)
        return launch_gemm_tn128x256p<IS_F16>(st, A, lda, W, ldw, M, N, K, epi, force == 20 ? 1 : 0, gp_dephase_cycles(K));
    if (force == 20 || force == 21) force = 6;
#else       // measured and rejected (DESIGN.md §4 "Round 3"): not part of the product library
    if (force == 20 || force == 21)
        return fail(VQ_ERR_INVALID, "gemm kernel %d is an experiment: rebuild with `make EXPERIMENTS=1`", force);
#endif
    // 12: 128x256 tiles, two workgroups per CU, wherever the 256x256 kernel would run (13: on every shape that tiles)
#ifdef VQ_GEMM_EXPERIMENTS
    if ((force == 12 || force == 13) && M % G12_BM == 0 && N % G12_BN == 0 && K % (2 * G12_SUB_K) == 0 && K >= 4 * G12_SUB_K &&
        (force == 13 || (int64_t)(M / G12_BM) * (N / G12_BN) >= 256))
        return launch_gemm_tn128x256<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
    if (force == 12 || force == 13) force = 6;
#else       // measured and rejected mainloops are not part of the product library
    if (force == 9 || force == 12 || force == 13)
        return fail(VQ_ERR_INVALID, "gemm kernel %d is an experiment: rebuild with `make EXPERIMENTS=1`", force);
#endif
    if constexpr (epi_row_in<Epi>::value) {
        // epilogues that consume per-row LayerNorm statistics need the kernels with the row-stat prologue
        const bool fits = M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0;
        const int64_t tiles = (int64_t)(M / G2_BM) * (N / G2_BN);
        if (fits && force != 1 && (force == 8 || tiles >= 128)) {
            const int rem = (int)(tiles % 256);
            if (force == 0 && gemm_use_tail_split() && tiles > 256 && rem > 0 && rem < 128) {      // thin last round -> 128x128 tiles
                const int m_main = (int)((tiles - rem) / (N / G2_BN)) * G2_BM;
                if (m_main > 0 && m_main < M) {
                    VQ_TRY((launch_gemm_tn256d<IS_F16>(st, A, lda, W, ldw, m_main, N, K, epi)));
                    return launch_gemm_tn<IS_F16>(st, A, lda, W, ldw, M - m_main, N, K, epi, m_main);
                }
            }
#ifdef VQ_GEMM_EXPERIMENTS
            if (force == 9 && lda % 64 == 0 && ldw % 64 == 0) return launch_gemm_tn256e<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
#endif
            // Three tiles of a tile row per workgroup where that still leaves >= 128 workgroups (fc1 and, since round 3, qkv at batch
            // 256): the second and third tile's first operands land under the previous epilogue (fc1 -3.5 % with one batch in flight,
            // +0.5 % frames/s with three; qkv drops to 150 workgroups: -29 % alone, +0.6-0.9 % frames/s with three batches in flight —
            // the idle CUs belong to the other batches then).  Concurrent handles only: a lone batch keeps the
            // tail-split dispatch below.  $VQ_AMD_GEMM_MULTI=0 switches it off, VQ_AMD_GEMM=15 forces it everywhere.
            if ((((force == 6 || force == 14) && gemm_use_multi() && tiles / 3 >= gemm_multi_min_wgs()) || force == 15) && lda % 64 == 0 && ldw % 64 == 0 && (N / G2_BN) % 3 == 0) {
                // [r04] FOUR tiles per workgroup where that fills the chip's 256 CUs better than three: ViT-L/14@336's q|k|v GEMM is 73 x 12
                // tiles = 292 workgroups of three (two rounds, the second 14 % full) or 219 of four (one round, 86 % full).  $VQ_AMD_GEMM_TPW forces.
                static const int tpw_env = [] { const char* e = getenv("VQ_AMD_GEMM_TPW"); return e ? atoi(e) : 0; }();
                auto fill = [&](int t) { const int64_t w = tiles / t; return (double)w / (double)(((w + 255) / 256) * 256); };
                int tpw = 3;
                if ((N / G2_BN) % 4 == 0 && tiles / 4 >= gemm_multi_min_wgs() && fill(4) > fill(3) + 0.05) tpw = 4;
                if (tpw_env >= 1 && (N / G2_BN) % tpw_env == 0) tpw = tpw_env;
                return launch_gemm_tn256dm<IS_F16>(st, A, lda, W, ldw, M, N, K, epi, tpw);
            }
            return launch_gemm_tn256d<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
        }
        return launch_gemm_tn<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
    } else {
    const bool allow160 = force != 6 && force != 14 && force != 15;
    if (force == 6) force = 0;
    if (force == 5 || (force == 0 && allow160 && gemm_use160() && prefer_tn160(M, N, K)))
        return launch_gemm_tn160_ring<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
#ifdef VQ_GEMM_EXPERIMENTS
    if (force == 7) return launch_gemm_tn256w4<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
#else
    if (force == 3 || force == 4 || force == 7)
        return fail(VQ_ERR_INVALID, "gemm kernel %d is an experiment: rebuild with `make EXPERIMENTS=1`", force);
#endif
    const bool fits256 = M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0;
    if (force == 8) return launch_gemm_tn256d<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
    if (force == 11) return launch_gemm_tn256d<IS_F16, Epi, false>(st, A, lda, W, ldw, M, N, K, epi);
#ifdef VQ_GEMM_EXPERIMENTS
    if (force == 9 && fits256 && lda % 64 == 0 && ldw % 64 == 0) return launch_gemm_tn256e<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
#endif
    if (force == 15 && fits256 && lda % 64 == 0 && ldw % 64 == 0 && (N / G2_BN) % 3 == 0 && (int64_t)(M / G2_BM) * (N / G2_BN) >= 128)
        return launch_gemm_tn256dm<IS_F16>(st, A, lda, W, ldw, M, N, K, epi, 3);
    if (force == 16 && fits256 && lda % 64 == 0 && ldw % 64 == 0)          // tests: the multi-tile kernel on any shape that tiles
        return launch_gemm_tn256dm<IS_F16>(st, A, lda, W, ldw, M, N, K, epi, (N / G2_BN) % 3 == 0 ? 3 : (N / G2_BN) % 4 == 0 ? 4 : (N / G2_BN) % 2 == 0 ? 2 : 1);
    if (force == 14 || force == 15 || force == 16) force = 0;
#ifdef VQ_GEMM_EXPERIMENTS
    if (force == 10) return launch_gemm_tn256f<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
#else
    if (force == 10)
        return fail(VQ_ERR_INVALID, "gemm kernel %d is an experiment: rebuild with `make EXPERIMENTS=1`", force);
#endif
    if (force == 2) return launch_gemm_tn256<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
    const bool want256 = force >= 2 || (force == 0 && (int64_t)(M / G2_BM) * (N / G2_BN) >= 128);
#ifdef VQ_GEMM_EXPERIMENTS
    if (fits256 && want256 && force == 4) return launch_gemm_tn256p<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
    if (fits256 && want256 && force == 3) return launch_gemm_tn256_ring<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
#endif
    if (fits256 && want256 && force == 0 && allow160 && gemm_use_tail_split()) {
        // Tile quantisation: T tiles over 256 CUs run ceil(T/256) rounds.  When the last round is less than half
        // full, its tiles' rows go to the 128x128 kernel instead (4x the workgroups, two per CU: one short round)
        // — same K order per output element, so the results are bit-identical.  Not for concurrent handles:
        // other streams fill the idle CUs of a thin round.
        const int tiles_n = N / G2_BN, tiles_m = M / G2_BM;
        const int64_t tiles = (int64_t)tiles_m * tiles_n;
        const int rem = (int)(tiles % 256);
        if (tiles > 256 && rem > 0 && rem < 128) {
            const int main_rows_tiles = (int)((tiles - rem) / tiles_n);          // whole tile rows inside the full rounds
            const int m_main = main_rows_tiles * G2_BM;
            if (m_main > 0 && m_main < M) {
                VQ_TRY((launch_gemm_tn256_best<IS_F16>(st, A, lda, W, ldw, m_main, N, K, epi)));
                return launch_gemm_tn<IS_F16>(st, A, lda, W, ldw, M - m_main, N, K, epi, m_main);
            }
        }
    }
    if (fits256 && want256 && force != 1) return launch_gemm_tn256_best<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
    return launch_gemm_tn<IS_F16>(st, A, lda, W, ldw, M, N, K, epi);
    }
}

}  // namespace vq

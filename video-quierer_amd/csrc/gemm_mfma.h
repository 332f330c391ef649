// bf16/fp16 "TN" GEMM on gfx950 MFMA with a fused epilogue functor.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        A: [M][K] row-major (activations)
//                                            W: [N][K] row-major (nn.Linear layout [out,in])
//
// Both operands are K-contiguous, so both MFMA fragments are plain 16-byte LDS
// reads.  The weight fragment is fed as the MFMA "A" operand and the activation
// fragment as "B": the 16x16 result then has n on the register axis, i.e. each
// lane owns 4 CONSECUTIVE n for one m, which makes the epilogue's global
// accesses 8-byte (bf16) / 16-byte (fp32) wide instead of 2-byte scatters.
//
// Tile: 128(m) x 128(n) x 64(k) per 256-thread workgroup, 4 waves as 2x2, each
// wave 64x64 = 4x4 MFMA 16x16x32 tiles.  Staging is LDS-DMA
// (global_load_lds, 16 B/lane): one wave-instruction fills 8 rows x 128 B; the
// 16-byte chunk index is XOR-swizzled with (row>>1)&7 on the SOURCE side (the
// LDS image itself is lane-linear) and the same XOR is applied on the fragment
// read, which makes every ds_read_b128 lane group hit 16 distinct 16-B slots of
// the 256-B bank row (conflict-free; checked with SQ_LDS_BANK_CONFLICT).
// Double-buffered: the next k-tile's DMA is issued before the MFMAs of the
// current one.
//
// Requirements (checked by the host launcher): M % 128 == 0 (buffers are padded),
// N % 128 == 0, K % 64 == 0, lda/ldw multiples of 8 elements, 16-B aligned bases.
#pragma once
#include "vq_common.h"
#include <type_traits>

namespace vq {

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64;
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_LDS_BYTES = 2 * (GEMM_BM + GEMM_BN) * GEMM_BK * 2;   // 64 KiB

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <bool IS_F16> struct mfma_op;
template <> struct mfma_op<false> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct mfma_op<true> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 run(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// XCD-aware bijective remap of the flat workgroup id (blocks b and b+8 share an
// XCD's L2): each XCD gets a contiguous run of tiles, n fastest, so the A panel
// of a tile row is re-read from that XCD's L2 by the N/128 tiles that share it.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// logical tile index (0 .. tiles_m*tiles_n-1, no padding) -> (tile_m, tile_n) in 2-D blocked order:
// m-block rows of bm tile rows; inside a row, n-blocks of bn tile columns; inside a block, n fastest.
__device__ __forceinline__ bool tile_coords(int idx, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int bn = tiles_n < 8 ? tiles_n : 8;
    const int bm = 32 / bn > 0 ? 32 / bn : 1;
    const int row_tiles = bm * tiles_n;
    const int bmi = idx / row_tiles, rem = idx - bmi * row_tiles;
    const int bm_eff = min(bm, tiles_m - bmi * bm);
    const int blk_tiles = bm_eff * bn;
    const int bni = rem / blk_tiles, rem2 = rem - bni * blk_tiles;
    const int bn_eff = min(bn, tiles_n - bni * bn);
    tm = bmi * bm + rem2 / bn_eff;
    tn = bni * bn + rem2 % bn_eff;
    return true;
}
// Epilogue staging.  The MFMA result layout gives a lane 4 consecutive n for ONE m, so a wave-wide
// store touches 16 different rows with 64-byte pieces; measured on the 256x256 kernel that store
// pattern alone cost ~30-40 us per launch (39 MB of fp32 at <1 TB/s).  Instead every wave transposes
// its accumulators through a private LDS strip (32 rows x 64 cols fp32, row stride 272 B: conflict-free
// for the b128 write and read lane groups) and calls the epilogue functor with ROW-contiguous data:
// 16 lanes x 16 B = one 256-byte row segment, 4 rows per wave instruction.
constexpr int EPI_ROW_BYTES = 272;
constexpr int EPI_WAVE_BYTES = 32 * EPI_ROW_BYTES;      // 8704 B per wave

// Epilogue functor interface (row-contiguous calls; n is the same for every call of a lane):
//   f32x4 bias_at(n)                  loaded once per lane
//   static constexpr bool kLoads      whether the epilogue reads memory per element (residual RMW, position emb.)
//   f32x4 load(m, n)                  that read; all 8 reads of a pass are issued BEFORE the first store, because a
//                                     load->add->store chain per element serialises on the L2 round trip (measured:
//                                     32 dependent round trips per lane in the residual GEMMs)
//   void  store(m, n, acc, bias, loaded)
// Optional functor features (LayerNorm folded into the GEMMs around it, encoder_kernels.h):
//   kRowStats  the functor's store_stats() returns the values it stored (the new residual row segment); the epilogue
//              sums them and their squares over the wave's 64 columns and hands the row's partial (sum, sum of squares)
//              to put_stats(m, n_wave0, s1, s2) — one partial per (row, 64-column granule), no atomics
//   kWide      (with kRowIn, 16-bit outputs) a lane owns EIGHT consecutive columns of a row and stores them as one 16-byte
//              piece, 8 rows per wave instruction: store_ln8(m, n, v0, v1, bias0, bias1, aux0, aux1, stat).  Half as many
//              store instructions and loop trips as the 4-column form; 8-byte stores per lane are issue-bound
//              (MI355X_MICROARCH.md, 'attention epilogue store tail': 8x dwordx4 halves 16x dwordx2)
//   kWideRes   [r04] the 8-column form for a residual epilogue that reads memory and emits row statistics (the 16 + 16-bit residual
//              stream: four 8-byte accesses per four elements would be issue-bound): load8(m, n8, A, B) fetches the row segment as
//              two 16-byte pieces one pass ahead, store_stats8(m, n8, v0, v1, bias0, bias1, A, B, s1, s2) stores and returns the
//              lane's sum and sum of squares; the 8 lanes of a row segment are reduced by DPP adds (row8_sum)
//   kRowIn     the functor consumes per-row (mean, rstd) prepared by the kernel prologue (row_stat(m), an LDS read) and a
//              second per-column constant aux_at(n); its store is store_ln(m, n, acc, bias, aux, stat)
template <class E, class = void> struct epi_row_stats : std::false_type {};
template <class E> struct epi_row_stats<E, std::void_t<decltype(E::kRowStats)>> : std::bool_constant<E::kRowStats> {};
template <class E, class = void> struct epi_split_k : std::false_type {};
template <class E> struct epi_split_k<E, std::void_t<decltype(E::kSplitK)>> : std::bool_constant<E::kSplitK> {};
template <class E, class = void> struct epi_wide : std::false_type {};
template <class E> struct epi_wide<E, std::void_t<decltype(E::kWide)>> : std::bool_constant<E::kWide> {};
template <class E, class = void> struct epi_wide_res : std::false_type {};
template <class E> struct epi_wide_res<E, std::void_t<decltype(E::kWideRes)>> : std::bool_constant<E::kWideRes> {};
template <class E, class = void> struct epi_row_in : std::false_type {};
template <class E> struct epi_row_in<E, std::void_t<decltype(E::kRowIn)>> : std::bool_constant<E::kRowIn> {};

template <class Epi>
__device__ __forceinline__ void epi_emit(const Epi& epi, int m, int n, int n_wave0, int rcol, f32x4 v, f32x4 bias, f32x4 aux, f32x4 loaded) {
    if constexpr (epi_row_in<Epi>::value) {
        epi.store_ln(m, n, v, bias, aux, epi.row_stat(m));
    } else if constexpr (epi_row_stats<Epi>::value) {
        const f32x4 r = epi.store_stats(m, n, v, bias, loaded);
        float s1 = (r[0] + r[1]) + (r[2] + r[3]);
        float s2 = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
        s1 = row16_sum(s1); s2 = row16_sum(s2);                  // the 16 lanes of a row: fixed order (vq_common.h)
        if (rcol == 0) epi.put_stats(m, n_wave0, s1, s2);
    } else {
        epi.store(m, n, v, bias, loaded);
    }
}

// kRowIn functors: the workgroup computes (mean, rstd) of its BM tile rows into LDS once (while its first operand
// tiles are in flight) and binds a private copy of the functor to them.
template <int BM, class Epi>
__device__ __forceinline__ Epi epi_bind_rowstats(const Epi& epi, float2* lds_stats, int m0, int tid, int nthreads) {
    Epi e = epi;
    if constexpr (epi_row_in<Epi>::value) {
        for (int r = tid; r < BM; r += nthreads) lds_stats[r] = epi.make_row_stat(m0 + r);
        e.lds = lds_stats; e.m0 = m0;
    }
    return e;
}

// C = acc, fp32 (unit-test hook vq_debug_gemm and the diagnostic timers)
struct EpiStoreF32 {
    float* out; int ldo;
    static constexpr bool kLoads = false;
    __device__ __forceinline__ f32x4 bias_at(int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 load(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void store(int m, int n, f32x4 v, f32x4, f32x4) const {
        *(f32x4*)(out + (size_t)m * ldo + n) = v;
    }
};

struct EpiNoHook { __device__ __forceinline__ void operator()() const {} };

// `after_loads()` runs once, right behind the epilogue's first global loads (per-column constants, first residual rows):
// the multi-tile kernel issues the NEXT tile's LDS-DMA there — vmcnt retires in order, so a load issued behind that DMA
// would wait for it to land, which is the very latency the DMA is issued early to hide.
template <int MI, class Epi, class Hook = EpiNoHook>      // wave tile = MI*16 rows x 64 cols; acc[mi][ni] = C[16mi + lane&15][16ni + 4(lane>>4) ..+3]
__device__ __forceinline__ void wave_epilogue(char* strip, const f32x4 (&acc)[MI][4], int m_wave0, int n_wave0,
                                              int lane, const Epi& epi, const Hook& after_loads = Hook()) {
    const int frow = lane & 15, fgrp = lane >> 4;
    if constexpr (epi_wide<Epi>::value) {
        static_assert(epi_row_in<Epi>::value && !Epi::kLoads, "the 8-column epilogue form is for the LayerNorm-consuming 16-bit outputs");
        const int wrow = lane >> 3, wcol = lane & 7;
        const int n8 = n_wave0 + wcol * 8;
        const f32x4 b0 = epi.bias_at(n8), b1 = epi.bias_at(n8 + 4), a0 = epi.aux_at(n8), a1 = epi.aux_at(n8 + 4);
        after_loads();
#pragma unroll
        for (int pass = 0; pass < (MI + 1) / 2; ++pass) {
            const int blocks = (2 * pass + 1 < MI) ? 2 : 1;               // 16-row blocks in this pass (an odd MI ends on one)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (h < blocks) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        *(f32x4*)(strip + (h * 16 + frow) * EPI_ROW_BYTES + (ni * 16 + fgrp * 4) * 4) = acc[pass * 2 + h][ni];
                }
            // same wave, in-order LDS: the reads below see the writes above
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (it < 2 * blocks) {
                    const int row = it * 8 + wrow;
                    const f32x4 v0 = *(const f32x4*)(strip + row * EPI_ROW_BYTES + wcol * 32);
                    const f32x4 v1 = *(const f32x4*)(strip + row * EPI_ROW_BYTES + wcol * 32 + 16);
                    const int m = m_wave0 + pass * 32 + row;
                    epi.store_ln8(m, n8, v0, v1, b0, b1, a0, a1, epi.row_stat(m));
                }
            }
        }
        return;
    }
    if constexpr (epi_wide_res<Epi>::value) {
        const int wrow = lane >> 3, wcol = lane & 7;
        const int n8 = n_wave0 + wcol * 8;
        const f32x4 b0 = epi.bias_at(n8), b1 = epi.bias_at(n8 + 4);
        constexpr int FULLW = MI / 2;                                     // passes over 32 rows = 4 wave instructions of 8 rows
        uint4 ldA[2][4], ldB[2][4];                                       // the row segments of a pass, fetched one pass ahead
#pragma unroll
        for (int it = 0; it < (FULLW > 0 ? 4 : 2); ++it) epi.load8(m_wave0 + it * 8 + wrow, n8, ldA[0][it], ldB[0][it]);
        after_loads();
#pragma unroll
        for (int pass = 0; pass < (MI + 1) / 2; ++pass) {
            const int blocks = (2 * pass + 1 < MI) ? 2 : 1;               // 16-row blocks in this pass (an odd MI ends on one)
            if (pass + 1 < (MI + 1) / 2) {
                const int nb = (2 * (pass + 1) + 1 < MI) ? 2 : 1;
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    if (it < 2 * nb) epi.load8(m_wave0 + (pass + 1) * 32 + it * 8 + wrow, n8, ldA[(pass + 1) & 1][it], ldB[(pass + 1) & 1][it]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (h < blocks) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        *(f32x4*)(strip + (h * 16 + frow) * EPI_ROW_BYTES + (ni * 16 + fgrp * 4) * 4) = acc[pass * 2 + h][ni];
                }
            // same wave, in-order LDS: the reads below see the writes above
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (it < 2 * blocks) {
                    const int row = it * 8 + wrow;
                    const f32x4 v0 = *(const f32x4*)(strip + row * EPI_ROW_BYTES + wcol * 32);
                    const f32x4 v1 = *(const f32x4*)(strip + row * EPI_ROW_BYTES + wcol * 32 + 16);
                    const int m = m_wave0 + pass * 32 + row;
                    float s1, s2;
                    epi.store_stats8(m, n8, v0, v1, b0, b1, ldA[pass & 1][it], ldB[pass & 1][it], s1, s2);
                    s1 = row8_sum(s1); s2 = row8_sum(s2);                 // the 8 lanes of the row segment: fixed order
                    if (wcol == 0) epi.put_stats(m, n_wave0, s1, s2);
                }
            }
        }
        return;
    }
    const int rrow = lane >> 4, rcol = lane & 15;
    const int n = n_wave0 + rcol * 4;
    const f32x4 bias = epi.bias_at(n);
    f32x4 aux = {0.f, 0.f, 0.f, 0.f};
    if constexpr (epi_row_in<Epi>::value) aux = epi.aux_at(n);
    constexpr int FULL = MI / 2;          // passes over 32 rows; an odd MI adds a last pass over 16 rows
    // Epilogues that read memory (residual RMW, position embedding) keep the NEXT pass's reads in flight while the
    // current pass goes through LDS and out: one exposed round trip per tile instead of one per pass (the residual
    // stream is 39 MB at batch 256, i.e. every read comes from beyond L2).
    f32x4 loaded[2][8];
    if constexpr (Epi::kLoads) {
#pragma unroll
        for (int it = 0; it < (FULL > 0 ? 8 : 4); ++it) loaded[0][it] = epi.load(m_wave0 + it * 4 + rrow, n);
    }
    after_loads();
#pragma unroll
    for (int pass = 0; pass < FULL; ++pass) {
        if constexpr (Epi::kLoads) {
            if (pass + 1 < FULL) {
#pragma unroll
                for (int it = 0; it < 8; ++it) loaded[(pass + 1) & 1][it] = epi.load(m_wave0 + (pass + 1) * 32 + it * 4 + rrow, n);
            } else if (MI & 1) {
#pragma unroll
                for (int it = 0; it < 4; ++it) loaded[(pass + 1) & 1][it] = epi.load(m_wave0 + (MI - 1) * 16 + it * 4 + rrow, n);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *(f32x4*)(strip + (h * 16 + frow) * EPI_ROW_BYTES + (ni * 16 + fgrp * 4) * 4) = acc[pass * 2 + h][ni];
        // same wave, in-order LDS: the reads below see the writes above
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + rrow;
            const f32x4 v = *(const f32x4*)(strip + row * EPI_ROW_BYTES + rcol * 16);
            epi_emit(epi, m_wave0 + pass * 32 + row, n, n_wave0, rcol, v, bias, aux, Epi::kLoads ? loaded[pass & 1][it] : f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
    if constexpr (MI & 1) {               // odd block count: a last pass over 16 rows
        constexpr int base = (MI - 1) * 16;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            *(f32x4*)(strip + frow * EPI_ROW_BYTES + (ni * 16 + fgrp * 4) * 4) = acc[MI - 1][ni];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 4 + rrow;
            const f32x4 v = *(const f32x4*)(strip + row * EPI_ROW_BYTES + rcol * 16);
            epi_emit(epi, m_wave0 + base + row, n, n_wave0, rcol, v, bias, aux, Epi::kLoads ? loaded[FULL & 1][it] : f32x4{0.f, 0.f, 0.f, 0.f});
        }
    }
}

template <bool IS_F16, class Epi>
__global__ __launch_bounds__(GEMM_THREADS, 2)
void gemm_tn_kernel(const uint16_t* __restrict__ A, int lda,
                    const uint16_t* __restrict__ W, int ldw,
                    int K, int tiles_n, Epi epi, int m_base, int tiles = 0 /* > 0: split-K — workgroup wg covers K-slice wg / tiles of tile wg % tiles (K = slice length) */) {
    typedef mfma_op<IS_F16> op;
    typedef typename op::frag frag;
    __shared__ __attribute__((aligned(16))) char smem[GEMM_LDS_BYTES + GEMM_BM * 8];     // + (mean, rstd) per tile row (kRowIn)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int wg = xcd_remap(blockIdx.x, gridDim.x);
    int kslice = 0;
    if (tiles > 0) { kslice = wg / tiles; wg -= kslice * tiles; A += (size_t)kslice * K; W += (size_t)kslice * K; }
    const int m0 = m_base + (wg / tiles_n) * GEMM_BM;        // m_base: first row of the strip this launch covers
    const int n0 = (wg % tiles_n) * GEMM_BN;

    // ---- staging addresses (per lane source, wave-uniform LDS destination) ----
    // wave w fills 1-KiB pieces 4w..4w+3 of the A tile and of the W tile.
    const int srow = lane >> 3;                  // row inside an 8-row piece
    const int sslot = lane & 7;                  // 16-B slot inside the 128-B row
    const uint16_t* a_src[4];
    const uint16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + srow;           // 0..127 inside the tile
        const int chunk = sslot ^ ((row >> 1) & 7);          // logical k-chunk this slot holds
        a_src[i] = A + (size_t)(m0 + row) * lda + chunk * 8;
        w_src[i] = W + (size_t)(n0 + row) * ldw + chunk * 8;
    }
    constexpr int TILE_BYTES = GEMM_BM * GEMM_BK * 2;        // 16 KiB per operand tile
    constexpr int BUF_BYTES = 2 * TILE_BYTES;

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * BUF_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_src[i] + kt * GEMM_BK),
                                             (lds_void_t*)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(w_src[i] + kt * GEMM_BK),
                                             (lds_void_t*)(base + TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets ----
    const int frow = lane & 15, fgrp = lane >> 4;
    const int fx = (frow >> 1) & 7;
    // byte offset of (tile-row block t, k-substep ks) for this lane:
    //   row = 16 t + frow ; slot = (4 ks + fgrp) ^ fx
    const int a_off = (wm * 64 + frow) * 128;
    const int w_off = TILE_BYTES + (wn * 64 + frow) * 128;
    const int slot0 = ((0 + fgrp) ^ fx) * 16, slot1 = ((4 + fgrp) ^ fx) * 16;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / GEMM_BK;
    stage(0, 0);
    Epi epi_wg = epi_bind_rowstats<GEMM_BM>(epi, (float2*)(smem + GEMM_LDS_BYTES), m0, tid, GEMM_THREADS);
    if constexpr (epi_split_k<Epi>::value) epi_wg.slice = kslice;
    __syncthreads();          // emits vmcnt(0): the DMA has landed for every wave

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* buf = smem + cur * BUF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = ks ? slot1 : slot0;
            frag af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const frag*)(buf + a_off + t * 2048 + so);
                wf[t] = *(const frag*)(buf + w_off + t * 2048 + so);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = op::run(wf[ni], af[mi], acc[mi][ni]);
        }
        __syncthreads();      // next buffer landed (vmcnt(0)) and this one is free to refill
    }

    // ---- epilogue (the last __syncthreads() retired every fragment read: LDS is free) ----
    wave_epilogue<4>(smem + wave * EPI_WAVE_BYTES, acc, m0 + wm * 64, n0 + wn * 64, lane, epi_wg);
}

// Split-K form for GEMMs with few output tiles and a long K (the CLS-only last block: 12 tiles x K = 3072): `splits`
// workgroups per tile, each over K / splits, results to the functor's per-slice partial planes (kSplitK functors);
// a reduce kernel (encoder_kernels.h) sums the planes in slice order, so the result does not depend on timing.
template <bool IS_F16, class Epi>
static int launch_gemm_tn_splitk(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                                 int M, int N, int K, int splits, const Epi& epi) {
    static_assert(epi_split_k<Epi>::value, "split-K needs a functor with per-slice outputs");
    VQ_CHECK(M > 0 && M % GEMM_BM == 0 && N % GEMM_BN == 0 && splits > 0 && K % (splits * GEMM_BK) == 0,
             "gemm_tn_splitk: shape M=%d N=%d K=%d / %d slices is not tile-aligned", M, N, K, splits);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn_splitk: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    const int tiles_m = M / GEMM_BM, tiles_n = N / GEMM_BN;
    hipLaunchKernelGGL((gemm_tn_kernel<IS_F16, Epi>), dim3(tiles_m * tiles_n * splits), dim3(GEMM_THREADS), 0, st,
                       A, lda, W, ldw, K / splits, tiles_n, epi, 0, tiles_m * tiles_n);
    VQ_HIP(hipGetLastError());
    return 0;
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                          int M, int N, int K, const Epi& epi, int m_base = 0) {
    VQ_CHECK(M > 0 && M % GEMM_BM == 0 && N % GEMM_BN == 0 && K % GEMM_BK == 0 && K >= GEMM_BK,
             "gemm_tn: shape M=%d N=%d K=%d is not tile-aligned (128/128/64)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    const int tiles_m = M / GEMM_BM, tiles_n = N / GEMM_BN;
    hipLaunchKernelGGL((gemm_tn_kernel<IS_F16, Epi>), dim3(tiles_m * tiles_n), dim3(GEMM_THREADS), 0, st,
                       A, lda, W, ldw, K, tiles_n, epi, m_base);
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace vq

// Small-batch scan: the reference's own call pattern is ONE query at a time, k*2 results
// (src/video_search_system.py:297 `index.search(query_vector, k * 2)`), i.e. a matrix-vector product: every
// fp16 row is read once from HBM per pass and does 2*Q FLOP per element — HBM-bound for Q below ~300
// (SURVEY.md §8d).  The 256-query MFMA tile of scan2_f16_top2_kernel wastes 255/256 of its matrix work there.
//
// This kernel streams the fp16 matrix straight into MFMA operand registers, no LDS:
//   one wave = one stream of 128 consecutive rows = 8 blocks of 16 rows; per block 16 wave-wide 16-byte loads
//   (lane l: row l&15, halves 32*ks + 8*(l>>4) ..+7 — exactly the A fragment of v_mfma_f32_16x16x32_f16), the
//   next block's 16 KiB in flight while the current one is multiplied; up to 16 queries are the B operand and
//   live in registers for the whole kernel (dim/32 fragments).  D[row][query]: lane (q = l&15, g = l>>4) gets
//   rows 4g..4g+3 of the block; it folds its 32 scores per stream into a top-2 (score with the 7-bit local row
//   index in the low mantissa bits, as scan*_f16_top2 do), the four lanes of a query merge their top-2 by
//   shuffles, and lane g = 0 writes the stream's two keys.  Key layout 3 (rescore_verify_small_kernel): QUERY-major
//   [q_pad][streams][2], row = stream*128 + local — the single query's re-score pass reads its 62 KB of keys as one
//   contiguous run (stream-major, they lay 128 B apart over 1 MB: sixteen 64-KB translations and 64 lines per wave load,
//   20k cycles until the first 16 keys per thread had landed).
//
// NQG = 2 keeps TWO groups of 16 queries in registers (128 VGPRs of B fragments at dim 512), so 17..32 queries still read
// the matrix ONCE (q32 was two passes = 2 GB for 0.365 ms; the MFMA and fold work per 16 KiB block stays far under the
// HBM time).  To make room the row fragments are single-buffered and refilled in rotation: fragment ks of the NEXT block is
// requested right after the MFMAs that consumed fragment ks of this one, so 16 loads (16 KiB per wave) stay in flight.
// More queries: one pass per 16*NQG (blockIdx.y), Q <= SCAN3_MAX_Q.
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"

namespace vq {

constexpr int SCAN3_QB = 16;            // queries per pass
constexpr int SCAN3_FUSED_MAX_Q = 4;    // up to this many queries the scan converts them itself (template FUSED)
constexpr int SCAN3_MAX_Q = 96;         // three 32-query passes (0.22 ms each over 1M x 512) still beat one 256-query MFMA tile (0.83 ms)

__host__ __device__ inline int64_t scan3_row_of(int64_t stream, int local) { return stream * 128 + local; }

// FUSED (one to four queries — the reference's own call is ONE): the queries arrive as fp32 [nq_real][dim] and are rounded to
// fp16 here, by the lanes that hold a real query; the other lanes of the B operand are zero without a load, and only real
// queries' keys are written.  That removes the conversion launch in front of a single-query search (~7 us of its ~210), the
// 16 KiB of (mostly zero) fp16 query rows every one of the 31k waves used to fetch (0.5 GB of L2 reads beside the 1 GB matrix
// stream) and 15 of 16 scattered 8-byte key stores per wave.  Same rounding (_Float16 cast = v_cvt_f16_f32, RNE) as
// queries_to_f16_kernel: keys bit-identical to the two-launch path.
template <int NKS, int NQG, bool FUSED = false>             // dim / 32; groups of 16 queries per pass (1 or 2)
__global__ __launch_bounds__(256, 2)
void scan3_f16_top2_kernel(const uint16_t* __restrict__ Q16 /*[q_pad][dim]; FUSED: const float* [nq_real][dim]*/, const uint16_t* __restrict__ X16,
                           int64_t n_valid, int64_t streams, int64_t q_pad /* % (16 NQG) == 0 */,
                           uint32_t* __restrict__ keys /*[q_pad][streams][2]*/, int nq_real = 0 /* FUSED only */) {
    typedef mfma_op<true> op;
    typedef op::frag frag;
    constexpr int DIM = NKS * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t stream = (int64_t)blockIdx.x * 4 + wave;
    if (stream >= streams) return;                         // wave-uniform; no barriers below
    const int r16 = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.y * SCAN3_QB * NQG;

    frag qf[NQG][NKS];
    if constexpr (FUSED) {
        static_assert(NQG == 1, "the fused form serves up to four queries: one group");
        const float* qsrc = (const float*)Q16 + (size_t)(r16 < nq_real ? r16 : 0) * DIM + g * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            frag f = {0, 0, 0, 0, 0, 0, 0, 0};
            if (r16 < nq_real) {
                const float4 a = *(const float4*)(qsrc + ks * 32), b = *(const float4*)(qsrc + ks * 32 + 4);
                f = frag{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
            }
            qf[0][ks] = f;
        }
    } else {
#pragma unroll
        for (int qg = 0; qg < NQG; ++qg)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                qf[qg][ks] = *(const frag*)(Q16 + (size_t)(q0 + qg * SCAN3_QB + r16) * DIM + ks * 32 + g * 8);
    }

    const uint16_t* xrow = X16 + ((size_t)stream * 128 + r16) * DIM + g * 8;
    frag xf[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) xf[ks] = *(const frag*)(xrow + ks * 32);

    const float NEG = -__builtin_inff(), MASKED = -3.0e38f;   // finite sentinel: see scan_f16_top2_kernel
    float m1[NQG], m2[NQG];
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg) { m1[qg] = NEG; m2[qg] = NEG; }
    const bool ragged = (stream + 1) * 128 > n_valid;      // wave-uniform
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
        f32x4 acc[NQG];
#pragma unroll
        for (int qg = 0; qg < NQG; ++qg) acc[qg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int qg = 0; qg < NQG; ++qg) acc[qg] = op::run(xf[ks], qf[qg][ks], acc[qg]);
            if (rb + 1 < 8) {                                                                            // refill in rotation
                xf[ks] = *(const frag*)(xrow + (size_t)(rb + 1) * 16 * DIM + ks * 32);
                __builtin_amdgcn_sched_barrier(0);      // or the scheduler sinks the load to its use (one register, vmcnt(0) per MFMA)
            }
        }
#pragma unroll
        for (int qg = 0; qg < NQG; ++qg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[qg][r];
                const int local = rb * 16 + 4 * g + r;
                if (ragged && stream * 128 + local >= n_valid) v = MASKED;
                const float kf = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, v) & ~127u) | (uint32_t)local);
                m2[qg] = __builtin_amdgcn_fmed3f(m1[qg], m2[qg], kf);
                m1[qg] = fmaxf(m1[qg], kf);
            }
        __builtin_amdgcn_sched_barrier(0);      // the fold stays with its block (deferred folds kept 8 blocks of scores alive: spills)
    }
    // merge the four row groups of a query (lanes l, l^16, l^32, l^48): top-2 of two sorted pairs
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float b1 = __shfl_xor(m1[qg], o), b2 = __shfl_xor(m2[qg], o);
            const float lo = fminf(m1[qg], b1);
            m1[qg] = fmaxf(m1[qg], b1);
            m2[qg] = fmaxf(lo, fmaxf(m2[qg], b2));
        }
        if (g == 0 && (!FUSED || r16 < nq_real))
            *(uint2*)(keys + ((size_t)(q0 + qg * SCAN3_QB + r16) * streams + stream) * 2) =
                uint2{__builtin_bit_cast(uint32_t, m1[qg]), __builtin_bit_cast(uint32_t, m2[qg])};
    }
}

}  // namespace vq

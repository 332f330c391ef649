// Device kernels of the CLIP ViT image-encoder forward pass other than the GEMM
// mainloop: GEMM epilogues, patch extraction, LayerNorm, single-tile attention,
// pooling head.  gfx950 only.  Row references are to SURVEY.md §8a (E-rows).
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"

namespace vq {

// ============================ GEMM epilogues =================================
// Each is called with (m, n, v) where v = C[m][n..n+3] (fp32 accumulators).

// F16 = false: bf16 GEMM operands (the BASELINE config); true: fp16 operands (same MFMA rate, 8x
// smaller rounding error; what ViT-L/14@336 is specified with).
template <bool F16>
__device__ __forceinline__ uint2 pack4_h(f32x4 v) {
    if constexpr (F16) {
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        h4 b = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        return __builtin_bit_cast(uint2, b);
    } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};   // v_cvt_pk_bf16_f32 (RNE)
        return __builtin_bit_cast(uint2, b);
    }
}
template <bool F16>
__host__ __device__ inline uint16_t to_h16(float f) {
    if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)f);
    else return f32_to_bf16_rne(f);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { const float4 t = *(const float4*)p; return f32x4{t.x, t.y, t.z, t.w}; }

// y = acc + bias  -> 16-bit                    (E6: fused q|k|v projection)
template <bool F16>
struct EpiBiasH16 {
    uint16_t* out; int ldo; const float* bias;
    static constexpr bool kLoads = false;
    __device__ __forceinline__ f32x4 bias_at(int n) const { return ld4(bias + n); }
    __device__ __forceinline__ f32x4 load(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void store(int m, int n, f32x4 v, f32x4 b, f32x4) const {
        *(uint2*)(out + (size_t)m * ldo + n) = pack4_h<F16>(v + b);
    }
};

// y = quick_gelu(acc + bias) -> 16-bit         (E7: fc1; x*sigmoid(1.702x))
template <bool F16>
struct EpiBiasQuickGeluH16 {
    uint16_t* out; int ldo; const float* bias;
    static constexpr bool kLoads = false;
    __device__ __forceinline__ f32x4 bias_at(int n) const { return ld4(bias + n); }
    __device__ __forceinline__ f32x4 load(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void store(int m, int n, f32x4 v, f32x4 b, f32x4) const {
        v = v + b;
        // x*sigmoid(1.702x) = x * rcp(1 + 2^(-1.702*log2(e)*x)): v_exp_f32 is a base-2 exponential and v_rcp_f32
        // replaces the IEEE division sequence (1 ulp each; the result is rounded to 16 bits right after)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v[i]));
        *(uint2*)(out + (size_t)m * ldo + n) = pack4_h<F16>(v);
    }
};

// x += acc + bias   (fp32 residual stream)      (E5: out_proj / fc2 + residual)
// SITE only names the call site (0 = out_proj, 1 = fc2): the two GEMMs then are distinct kernels in a
// rocprofv3 --kernel-trace --stats summary instead of one line mixing 27-us and 80-us launches.
template <int SITE>
struct EpiBiasResidualF32 {
    float* x; int ldx; const float* bias;
    static constexpr bool kLoads = true;
    __device__ __forceinline__ f32x4 bias_at(int n) const { return ld4(bias + n); }
    __device__ __forceinline__ f32x4 load(int m, int n) const { return ld4(x + (size_t)m * ldx + n); }
    __device__ __forceinline__ void store(int m, int n, f32x4 v, f32x4 b, f32x4 r) const {
        *(f32x4*)(x + (size_t)m * ldx + n) = r + v + b;
    }
};

// Last block, CLS rows only: GEMM row m = image m -> residual row m*tokens; rows >= m_valid are padding
struct EpiBiasResidualClsF32 {
    float* x; int hidden; int tokens; const float* bias; int m_valid;
    static constexpr bool kLoads = true;
    __device__ __forceinline__ f32x4 bias_at(int n) const { return ld4(bias + n); }
    __device__ __forceinline__ f32x4 load(int m, int n) const {
        return m < m_valid ? ld4(x + (size_t)m * tokens * hidden + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void store(int m, int n, f32x4 v, f32x4 b, f32x4 r) const {
        if (m < m_valid) *(f32x4*)(x + (size_t)m * tokens * hidden + n) = r + v + b;
    }
};

// ---- LayerNorm folded into the GEMMs around it ---------------------------------------------------------------
// LN(x)[m][k] = (x[m][k] - mu_m) rstd_m g[k] + b[k], so for the GEMM that consumes it
//     sum_k LN(x)[m][k] W[n][k] = rstd_m (sum_k x[m][k] W'[n][k]  -  mu_m c1[n])  +  c2[n]
// with W'[n][k] = g[k] W[n][k] (folded on the host, then rounded to 16 bits), c1[n] = sum_k W'[n][k] (of the ROUNDED
// weights: what the MFMA really multiplies), c2[n] = sum_k b[k] W[n][k] + bias[n].  The GEMM's A operand is then the
// raw residual stream rounded to 16 bits (`xh`), which the PRODUCER of x writes next to the fp32 x — the residual
// epilogues below and the embedding kernels — together with per-row partial sums (sum x, sum x^2) per 64-column
// granule: ps[granule][row].  No separate LayerNorm pass reads x again (tf:362-383 pre-LN blocks; reference
// src/core/feature_extractor.py:154).  Rounding x instead of LN(x) to 16 bits has the same relative error per
// element while |mean| <~ std over the row (true of ViT residual streams, whose rows are dominated by a few large
// channels of either sign; measured by the layerwise parity tests).
constexpr int LN_MAX_GRANULES = 16;                     // hidden <= 1024
struct LnPartials { float2* ps; int64_t stride; };     // ps[g * stride + row], g = column / 64; LN_MAX_GRANULES planes, unused ones stay zero

// x += acc + bias (fp32), xh = 16-bit(x), row partials -> ps          (producer: out_proj / fc2)
//
// [r04] The residual stream as 16 + 16 bits.  xh = fp16(x) IS the next GEMM's operand and is written anyway; with
// xl = fp16(x - xh) beside it the pair carries x to ~22 bits (fp32 has 24), so the fp32 copy need not be written — and
// the next residual epilogue reads xh + xl (4 bytes per element, as before) instead of x.  Per element an epilogue then
// moves 4 B in + 4 B out instead of 4 B in + 6 B out: -20 % of the bytes of the two HBM-bound epilogues of the tower
// (out_proj: 130 -> 111 MB per launch for 15 GFLOP, fc2 222 -> 205 MB: PMC).  MODE bits: RS_IN_SPLIT (read xh + xl, else
// the fp32 x), RS_OUT_SPLIT (write xl), RS_OUT_F32 (write the fp32 x).  The first residual epilogue of a pass reads the fp32 x
// the embedding kernel wrote; the last one before a consumer of x itself (the pooling head, the CLS-only last block) writes it
// again.  fp16 operands only (F16): with bf16 operands xh has 8 bits and the pair 16.
// (v_fma_mix_f32 for hi + lo and y - hi — one instruction per element instead of conversions + add — was built and measured:
//  same time to 0.1 us, so the plain form stays.)
// Streaming (nt) stores for epilogue outputs that are far larger than the L2s and consumed by the NEXT kernel: the q|k|v and MLP
// activations (59 / 79 MB per launch).  Written through the L2 like ordinary data they displace the operand panels the GEMM is
// re-reading — PMC, one batch at a time: qkv 151.0 -> 119.6 MB from beyond L2 per launch, fc1 183.9 -> 165.5 MB; 64.9 -> 60.9 us
// and 74.9 -> 71.3 us; three batches in flight 100.9k -> 101.6-102.2k frames/s (profiles/r04_ab_nt_store/).  On the residual
// epilogues' xh / xl (read back soon) they change nothing.  Build-time switches so that variants can be A/B-ed as separate libraries.
#ifndef VQ_EPI_LN_NT_STORE
#define VQ_EPI_LN_NT_STORE 1
#endif
#ifndef VQ_EPI_RES_NT_STORE
#define VQ_EPI_RES_NT_STORE 0
#endif
__device__ __forceinline__ void st8_maybe_nt(uint16_t* p, uint2 v) {
#if VQ_EPI_RES_NT_STORE
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    __builtin_nontemporal_store(u32x2{v.x, v.y}, (u32x2*)p);
#else
    *(uint2*)p = v;
#endif
}
enum : int { RS_IN_SPLIT = 1, RS_OUT_SPLIT = 2, RS_OUT_F32 = 4, RS_F32 = RS_OUT_F32 };
// [r04, second session] The low half as ONE byte: xl is stored as fp8 (OCP e4m3 of xl * 512, clamped to the format's +-448 - an
// out-of-range conversion is a NaN) instead of fp16: 3 bytes per element in and 3 out of a residual epilogue instead of 4 + 4.  |xl| is
// at most half an fp16 ulp of x, so xl * 512 <= |x| / 4: nothing clamps below |x| = 1,792, and the pair carries x to ~15 bits
// (|error| <= 2^-15 |x| for |x| >= 2^-6; below that xl flushes to zero and the error is <= 2^-17).  That is 16 times finer than the
// rounding of xh that every GEMM applies to its operand anyway: against the fp32 stream the embeddings move by 7.5e-5 (the 16 + 16-bit
// pair: 7e-5) and the score error against the fp32 pipeline is unchanged (7.2e-5 / 7.4e-5).  103.7k -> 105.7k frames/s with three
// batches in flight, same box (profiles/r04_ab_resid_xl8.txt).  VQ_RESID_XL8=0 builds the 16 + 16-bit form (A/B switch).
#ifndef VQ_RESID_XL8
#define VQ_RESID_XL8 1
#endif
__device__ __forceinline__ f32x4 unpack4_fp8(uint32_t w) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    return f32x4{lo[0], lo[1], hi[0], hi[1]} * (1.0f / 512.0f);
}
__device__ __forceinline__ uint32_t pack4_fp8(f32x4 v) {
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_fmed3f(v[i] * 512.0f, -448.0f, 448.0f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w, true);
    return (uint32_t)w;
}
__device__ __forceinline__ f32x4 unpack4_f16(uint2 u) {
    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
    const h4 h = __builtin_bit_cast(h4, u);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
template <int SITE, bool F16, int MODE = RS_F32>
struct EpiBiasResidualLnF32 {
    float* x; int ldx; const float* bias; uint16_t* xh; LnPartials part; uint16_t* xl = nullptr;
    static_assert(MODE == RS_F32 || F16, "the 16 + 16-bit residual stream needs fp16 operands");
    static constexpr bool kLoads = true, kRowStats = true;
    __device__ __forceinline__ f32x4 bias_at(int n) const { return ld4(bias + n); }
    __device__ __forceinline__ f32x4 load(int m, int n) const {
        if constexpr (MODE & RS_IN_SPLIT) {
            const size_t o = (size_t)m * ldx + n;
#if VQ_RESID_XL8
            return unpack4_f16(*(const uint2*)(xh + o)) + unpack4_fp8(*(const uint32_t*)((const uint8_t*)xl + o));
#else
            return unpack4_f16(*(const uint2*)(xh + o)) + unpack4_f16(*(const uint2*)(xl + o));
#endif
        } else {
            return ld4(x + (size_t)m * ldx + n);
        }
    }
    __device__ __forceinline__ void store(int, int, f32x4, f32x4, f32x4) const {}
    __device__ __forceinline__ f32x4 store_stats(int m, int n, f32x4 v, f32x4 b, f32x4 r) const {
        const f32x4 y = r + v + b;
        const size_t o = (size_t)m * ldx + n;
        if constexpr (MODE & RS_OUT_F32) *(f32x4*)(x + o) = y;
        const uint2 hi = pack4_h<F16>(y);
        st8_maybe_nt(xh + o, hi);
#if VQ_RESID_XL8
        if constexpr (MODE & RS_OUT_SPLIT) *(uint32_t*)((uint8_t*)xl + o) = pack4_fp8(y - unpack4_f16(hi));
#else
        if constexpr (MODE & RS_OUT_SPLIT) st8_maybe_nt(xl + o, pack4_h<true>(y - unpack4_f16(hi)));
#endif
        return y;
    }
    __device__ __forceinline__ void put_stats(int m, int n_wave0, float s1, float s2) const {
        part.ps[(size_t)(n_wave0 >> 6) * part.stride + m] = float2{s1, s2};
    }
    // The 8-column form (gemm_mfma.h kWideRes) for the modes that touch the 16 + 16-bit stream: every access a 16-byte piece.
    static constexpr bool kWideRes = (MODE & (RS_IN_SPLIT | RS_OUT_SPLIT)) != 0;
    __device__ __forceinline__ void load8(int m, int n8, uint4& a, uint4& b) const {
        const size_t o = (size_t)m * ldx + n8;
#if VQ_RESID_XL8
        if constexpr (MODE & RS_IN_SPLIT) { a = *(const uint4*)(xh + o); const uint2 t = *(const uint2*)((const uint8_t*)xl + o); b = uint4{t.x, t.y, 0u, 0u}; }
#else
        if constexpr (MODE & RS_IN_SPLIT) { a = *(const uint4*)(xh + o); b = *(const uint4*)(xl + o); }     // hi | lo
#endif
        else { a = *(const uint4*)(x + o); b = *(const uint4*)(x + o + 4); }                                // the fp32 x
    }
    __device__ __forceinline__ void store_stats8(int m, int n8, f32x4 v0, f32x4 v1, f32x4 b0, f32x4 b1, uint4 a, uint4 b,
                                                 float& s1, float& s2) const {
        f32x4 r0, r1;
        if constexpr (MODE & RS_IN_SPLIT) {
#if VQ_RESID_XL8
            r0 = unpack4_f16(uint2{a.x, a.y}) + unpack4_fp8(b.x);
            r1 = unpack4_f16(uint2{a.z, a.w}) + unpack4_fp8(b.y);
#else
            r0 = unpack4_f16(uint2{a.x, a.y}) + unpack4_f16(uint2{b.x, b.y});
            r1 = unpack4_f16(uint2{a.z, a.w}) + unpack4_f16(uint2{b.z, b.w});
#endif
        } else {
            r0 = __builtin_bit_cast(f32x4, a); r1 = __builtin_bit_cast(f32x4, b);
        }
        const f32x4 y0 = r0 + v0 + b0, y1 = r1 + v1 + b1;
        const size_t o = (size_t)m * ldx + n8;
        if constexpr (MODE & RS_OUT_F32) { *(f32x4*)(x + o) = y0; *(f32x4*)(x + o + 4) = y1; }
        const uint2 h0 = pack4_h<F16>(y0), h1 = pack4_h<F16>(y1);
        *(uint4*)(xh + o) = uint4{h0.x, h0.y, h1.x, h1.y};
        if constexpr (MODE & RS_OUT_SPLIT) {
#if VQ_RESID_XL8
            *(uint2*)((uint8_t*)xl + o) = uint2{pack4_fp8(y0 - unpack4_f16(h0)), pack4_fp8(y1 - unpack4_f16(h1))};
#else
            const uint2 l0 = pack4_h<true>(y0 - unpack4_f16(h0)), l1 = pack4_h<true>(y1 - unpack4_f16(h1));
            *(uint4*)(xl + o) = uint4{l0.x, l0.y, l1.x, l1.y};
#endif
        }
        s1 = ((y0[0] + y0[1]) + (y0[2] + y0[3])) + ((y1[0] + y1[1]) + (y1[2] + y1[3]));
        s2 = ((y0[0] * y0[0] + y0[1] * y0[1]) + (y0[2] * y0[2] + y0[3] * y0[3])) + ((y1[0] * y1[0] + y1[1] * y1[1]) + (y1[2] * y1[2] + y1[3] * y1[3]));
    }
};

// Last block, CLS rows only: GEMM row m = image m -> residual row m*tokens; xh and the partials are COMPACT (row m)
template <bool F16>
struct EpiBiasResidualClsLnF32 {
    float* x; int hidden; int tokens; const float* bias; int m_valid; uint16_t* xh; LnPartials part;
    static constexpr bool kLoads = true, kRowStats = true;
    __device__ __forceinline__ f32x4 bias_at(int n) const { return ld4(bias + n); }
    __device__ __forceinline__ f32x4 load(int m, int n) const {
        return m < m_valid ? ld4(x + (size_t)m * tokens * hidden + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void store(int, int, f32x4, f32x4, f32x4) const {}
    __device__ __forceinline__ f32x4 store_stats(int m, int n, f32x4 v, f32x4 b, f32x4 r) const {
        const f32x4 y = m < m_valid ? r + v + b : f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < m_valid) *(f32x4*)(x + (size_t)m * tokens * hidden + n) = y;
        *(uint2*)(xh + (size_t)m * hidden + n) = pack4_h<F16>(y);           // padding rows: zeros (finite operands downstream)
        return y;
    }
    __device__ __forceinline__ void put_stats(int m, int n_wave0, float s1, float s2) const {
        part.ps[(size_t)(n_wave0 >> 6) * part.stride + m] = float2{s1, s2};
    }
};

// y = rstd (acc - mu c1) + c2  [quick_gelu]  -> 16-bit                  (consumer: qkv / fc1 on xh)
// quick_gelu(y) = y * sigmoid(1.702 y) = y / (1 + 2^(-c y)), c = 1.702 log2(e).  [r04] c is folded into the fc1 weights and fold
// constants on the host (vq_encoder.hip upload_layer: the epilogue's y IS c y, the exponent a sign flip: one multiply per element
// fewer in the one epilogue of the tower that is bound by vector issue) and 1 / c into the fc2 weights; the stored MLP activations are
// c times the reference's.  fc1 71.5 -> 70.4 us per launch, +0.1 ... +0.6 % frames/s same box; score error against the fp32 pipeline
// unchanged (1.1e-4 / 1.25e-4).  VQ_GELU_FOLD=0 builds the unfolded form (A/B switch).
#ifndef VQ_GELU_FOLD
#define VQ_GELU_FOLD 1
#endif
constexpr float QUICK_GELU_C = 2.4554669595930157f;
#if VQ_GELU_FOLD
#define VQ_GELU_ARG(y) (-(y))
#else
#define VQ_GELU_ARG(y) (-QUICK_GELU_C * (y))
#endif

template <bool F16, bool GELU>
struct EpiLnH16 {
    uint16_t* out; int ldo; const float* c2; const float* c1; LnPartials part; int granules; float inv_h, eps;
    const float2* lds = nullptr; int m0 = 0;            // bound per workgroup by the kernel (epi_bind_rowstats)
    static constexpr bool kLoads = false, kRowIn = true, kWide = true;
    static constexpr bool kNtStore = VQ_EPI_LN_NT_STORE == 1 || (VQ_EPI_LN_NT_STORE == 2 && GELU);      // 1: q|k|v and MLP outputs, 2: the MLP output only
    __device__ __forceinline__ f32x4 bias_at(int n) const { return ld4(c2 + n); }
    __device__ __forceinline__ f32x4 aux_at(int n) const { return ld4(c1 + n); }
    __device__ __forceinline__ f32x4 load(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void store(int, int, f32x4, f32x4, f32x4) const {}
    __device__ __forceinline__ float2 make_row_stat(int m) const {          // (mean, rstd) of row m from its partials
        // all LN_MAX_GRANULES loads are issued before the first add (the buffer always has that many granule planes,
        // the unused ones zero): a loop with a run-time bound costs one exposed round trip per granule here, because
        // the workgroup's LDS-DMA prologue is in flight and the compiler drains the queue at every use of a load
        float2 p[LN_MAX_GRANULES];
#pragma unroll
        for (int g = 0; g < LN_MAX_GRANULES; ++g) p[g] = part.ps[(size_t)g * part.stride + m];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int g = 0; g < LN_MAX_GRANULES; ++g) { s1 += p[g].x; s2 += p[g].y; }
        const float mean = s1 * inv_h;
        const float var = fmaxf(s2 * inv_h - mean * mean, 0.f);
        return float2{mean, rsqrtf(var + eps)};
    }
    __device__ __forceinline__ float2 row_stat(int m) const { return lds[m - m0]; }
    __device__ __forceinline__ void store_ln(int m, int n, f32x4 v, f32x4 b, f32x4 a, float2 st) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y = st.y * (v[i] - st.x * a[i]) + b[i];
            if constexpr (GELU) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(VQ_GELU_ARG(y)));
            v[i] = y;
        }
        if constexpr (kNtStore) { typedef __attribute__((ext_vector_type(2))) unsigned int u32x2; const uint2 pk = pack4_h<F16>(v); __builtin_nontemporal_store(u32x2{pk.x, pk.y}, (u32x2*)(out + (size_t)m * ldo + n)); }
        else *(uint2*)(out + (size_t)m * ldo + n) = pack4_h<F16>(v);
    }
    __device__ __forceinline__ void store_ln8(int m, int n, f32x4 v0, f32x4 v1, f32x4 b0, f32x4 b1, f32x4 a0, f32x4 a1, float2 st) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y0 = st.y * (v0[i] - st.x * a0[i]) + b0[i];
            float y1 = st.y * (v1[i] - st.x * a1[i]) + b1[i];
            if constexpr (GELU) {
                y0 = y0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(VQ_GELU_ARG(y0)));
                y1 = y1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(VQ_GELU_ARG(y1)));
            }
            v0[i] = y0; v1[i] = y1;
        }
        const uint2 lo = pack4_h<F16>(v0), hi = pack4_h<F16>(v1);
        if constexpr (kNtStore) {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            __builtin_nontemporal_store(u32x4{lo.x, lo.y, hi.x, hi.y}, (u32x4*)(out + (size_t)m * ldo + n));
        } else {
            *(uint4*)(out + (size_t)m * ldo + n) = uint4{lo.x, lo.y, hi.x, hi.y};
        }
    }
};

// Split-K partial planes: part[slice][m][n] = this K-slice's accumulators (gemm_mfma.h launch_gemm_tn_splitk)
struct EpiSplitKPartialF32 {
    float* part; int ld; int64_t plane; int slice = 0;
    static constexpr bool kLoads = false, kSplitK = true;
    __device__ __forceinline__ f32x4 bias_at(int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 load(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void store(int m, int n, f32x4 v, f32x4, f32x4) const {
        *(f32x4*)(part + (size_t)slice * plane + (size_t)m * ld + n) = v;
    }
};

// x[m*tokens][n] += bias[n] + sum over slices (in slice order) of part[s][m][n], m < m_valid     (last block's fc2, CLS rows)
inline __global__ __launch_bounds__(256)
void splitk_reduce_residual_cls_kernel(const float* __restrict__ part, int64_t plane, int splits, float* __restrict__ x,
                                       const float* __restrict__ bias, int hidden, int tokens, int m_valid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // one float4 of one CLS row
    const int per_row = hidden >> 2;
    if (i >= m_valid * per_row) return;
    const int m = i / per_row, n = (i - m * per_row) * 4;
    f32x4 acc = ld4(bias + n);
    for (int s = 0; s < splits; ++s) acc = acc + ld4(part + (size_t)s * plane + (size_t)m * hidden + n);
    float* xr = x + (size_t)m * tokens * hidden + n;
    *(f32x4*)xr = ld4(xr) + acc;
}

// Patch-embedding GEMM: GEMM row m = (image b, patch p) -> token row b*T + 1 + p;
// x = acc + folded_bias + position_embedding[1+p]          (E3)
// `unscale` = 2^-s undoes the power-of-two scale the host put on fp16 patch weights (vq_encoder.hip: W/(255 std) sits in
// fp16's subnormal range unscaled); a power of two, so acc * unscale is exact.  1 for bf16 weights.
struct EpiPatchEmbedF32 {
    float* x; int hidden; const float* bias; const float* pos; int patches; int tokens; int m_valid; float unscale = 1.0f;
    static constexpr bool kLoads = true;
    __device__ __forceinline__ f32x4 bias_at(int n) const { return ld4(bias + n); }
    __device__ __forceinline__ f32x4 load(int m, int n) const {
        if (m >= m_valid) return f32x4{0.f, 0.f, 0.f, 0.f};
        const int p = m % patches;
        return ld4(pos + (size_t)(1 + p) * hidden + n);
    }
    __device__ __forceinline__ void store(int m, int n, f32x4 v, f32x4 b, f32x4 pp) const {
        if (m >= m_valid) return;
        const int bimg = m / patches, p = m - bimg * patches;
        *(f32x4*)(x + ((size_t)bimg * tokens + 1 + p) * hidden + n) = v * unscale + b + pp;
    }
};

// ============================ patch extraction ===============================
// uint8 frames [n][S][S][3] -> bf16 GEMM operand [n*P][3*ps*ps], K ordered
// (c, ky, kx) like Conv2d's weight [hidden,3,ps,ps] (E3).  The value stored is
// (pixel - 128): exact in bf16; ToTensor's /255 and Normalize's (x-mean)/std
// (E1) are folded into the patch weights and a per-output bias on the host.
// swap_rb=1 reverses the channel axis (cv2.COLOR_BGR2RGB for ndarray input, E1).
// One thread handles 8 consecutive pixels of one image row (24 contiguous bytes)
// and writes three 16-byte runs, one per channel plane.  Requires ps % 8 == 0.
template <bool F16>
__global__ __launch_bounds__(256)
void patchify_u8_kernel(const uint8_t* __restrict__ frames, uint16_t* __restrict__ out,
                        int n, int S, int ps, int swap_rb) {
    const int runs_per_row = S / 8;
    const int64_t total = (int64_t)n * S * runs_per_row;
    const int grid = S / ps;
    const int Kp = 3 * ps * ps;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int xr = (int)(t % runs_per_row);
        const int64_t t2 = t / runs_per_row;
        const int y = (int)(t2 % S);
        const int b = (int)(t2 / S);
        const uint8_t* src = frames + (((int64_t)b * S + y) * S + xr * 8) * 3;   // 8-B aligned
        const uint2 r0 = *(const uint2*)(src), r1 = *(const uint2*)(src + 8), r2 = *(const uint2*)(src + 16);
        const uint32_t w[6] = {r0.x, r0.y, r1.x, r1.y, r2.x, r2.y};
        const int gy = y / ps, ky = y - gy * ps;
        const int x0 = xr * 8, gx = x0 / ps, kx = x0 - gx * ps;
        uint16_t* dst = out + ((size_t)b * grid * grid + gy * grid + gx) * Kp + ky * ps + kx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cs = swap_rb ? 2 - c : c;                 // source channel for model channel c
            uint16_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int byte = i * 3 + cs;
                const int px = (w[byte >> 2] >> ((byte & 3) * 8)) & 0xff;
                v[i] = to_h16<F16>((float)(px - 128));
            }
            uint4 o = {(uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                       (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16)};
            *(uint4*)(dst + (size_t)c * ps * ps) = o;
        }
    }
}

// ============================ LayerNorm ======================================
// One wave per row; the row lives in registers (HIDDEN/256 float4 per lane).
// Two-pass mean / biased variance in fp32, like nn.LayerNorm (E4/E5/E8).

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int NV>   // NV = hidden / 256
__device__ __forceinline__ void ln_row(float4 (&v)[NV], const float* __restrict__ g,
                                       const float* __restrict__ b, int lane, float eps, int hidden) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / (float)hidden;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)hidden + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 gg = *(const float4*)(g + (i * 64 + lane) * 4);
        const float4 bb = *(const float4*)(b + (i * 64 + lane) * 4);
        v[i].x = v[i].x * rstd * gg.x + bb.x; v[i].y = v[i].y * rstd * gg.y + bb.y;
        v[i].z = v[i].z * rstd * gg.z + bb.z; v[i].w = v[i].w * rstd * gg.w + bb.w;
    }
}

// h = LN(x) as 16-bit (LN1 / LN2 feeding the next GEMM).  in_row_stride (in rows of x) lets the last
// block normalise only the CLS rows (stride = tokens) into a compact h.
template <int NV, bool F16>
__global__ __launch_bounds__(256)
void layernorm_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ h,
                           const float* __restrict__ g, const float* __restrict__ b,
                           int rows, float eps, int in_row_stride = 1) {
    constexpr int H = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *(const float4*)(x + (size_t)row * in_row_stride * H + (i * 64 + lane) * 4);
    ln_row<NV>(v, g, b, lane, eps, H);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        *(uint2*)(h + (size_t)row * H + (i * 64 + lane) * 4) = pack4_h<F16>(f32x4{v[i].x, v[i].y, v[i].z, v[i].w});
}

// The producer side of the folded LayerNorm for kernels that hold a whole row in a wave: xh = 16-bit(x) and the
// row's partial (sum, sum of squares) per 64-column granule — lane (i, l) holds columns (64 i + l) * 4 .. +3, so the
// 16 lanes l>>4 == g of register i are granule 4 i + g.
template <int NV, bool F16>
__device__ __forceinline__ void emit_xh_and_partials(const float4 (&v)[NV], uint16_t* __restrict__ xh_row, LnPartials part,
                                                     int row, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        *(uint2*)(xh_row + (i * 64 + lane) * 4) = pack4_h<F16>(f32x4{v[i].x, v[i].y, v[i].z, v[i].w});
        float s1 = (v[i].x + v[i].y) + (v[i].z + v[i].w);
        float s2 = (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        s1 = row16_sum(s1); s2 = row16_sum(s2);
        if ((lane & 15) == 0) part.ps[(size_t)(i * 4 + (lane >> 4)) * part.stride + row] = float2{s1, s2};
    }
}

// Embedding finish (E3 tail + E4): token 0 of every image is class_embedding + position_embedding[0]; then
// x = pre_layrnorm(x) in place (fp32 residual stream).  LN1 of layer 0 is folded into the qkv GEMM (EpiLnH16): this
// kernel is the producer of its operand xh = 16-bit(x) and of the row partials.
template <int NV, bool F16>
__global__ __launch_bounds__(256)
void embed_finish_kernel(float* __restrict__ x, uint16_t* __restrict__ xh,
                         const float* __restrict__ cls, const float* __restrict__ pos0,
                         const float* __restrict__ g_pre, const float* __restrict__ b_pre,
                         LnPartials part, int rows, int tokens, float eps) {
    constexpr int H = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool is_cls = (row % tokens) == 0;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int o = (i * 64 + lane) * 4;
        if (is_cls) {
            const float4 c = *(const float4*)(cls + o), p = *(const float4*)(pos0 + o);
            v[i] = float4{c.x + p.x, c.y + p.y, c.z + p.z, c.w + p.w};
        } else {
            v[i] = *(const float4*)(x + (size_t)row * H + o);
        }
    }
    ln_row<NV>(v, g_pre, b_pre, lane, eps, H);
#pragma unroll
    for (int i = 0; i < NV; ++i) *(float4*)(x + (size_t)row * H + (i * 64 + lane) * 4) = v[i];
    emit_xh_and_partials<NV, F16>(v, xh + (size_t)row * H, part, row, lane);
}

// rows r*stride of a 16-bit [.][cols] matrix -> compact rows r (CLS gather for the last block)
inline __global__ __launch_bounds__(256)
void gather_rows_h16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int rows, int cols, int stride) {
    const int per_row = cols / 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * per_row; i += gridDim.x * blockDim.x) {
        const int r = i / per_row, c = i - r * per_row;
        *(uint4*)(dst + (size_t)r * cols + c * 8) = *(const uint4*)(src + (size_t)r * stride * cols + c * 8);
    }
}

// ---- text tower embedding (tf:222-256): x = token_embedding[id] + position_embedding[t]; xh, partials for the folded LN1 ----
template <int NV, bool F16>
__global__ __launch_bounds__(256)
void embed_tokens_kernel(const int* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                         float* __restrict__ x, uint16_t* __restrict__ xh, LnPartials part,
                         int rows, int seq, int vocab) {
    constexpr int H = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int t = row % seq;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int o = (i * 64 + lane) * 4;
        const float4 a = *(const float4*)(tok + (size_t)id * H + o), p = *(const float4*)(pos + (size_t)t * H + o);
        v[i] = float4{a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w};
        *(float4*)(x + (size_t)row * H + o) = v[i];
    }
    emit_xh_and_partials<NV, F16>(v, xh + (size_t)row * H, part, row, lane);
}

// pooling row of every sequence (tf:568-586): first position holding eos_token_id (position 0 if none);
// for the legacy eos_token_id == 2 checkpoints, the position of the largest id.
inline __global__ void eos_rows_kernel(const int* __restrict__ ids, int* __restrict__ row_index, int n, int seq, int eos_id) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int* p = ids + (size_t)s * seq;
    int pos = 0;
    if (eos_id == 2) {
        int best = p[0];
        for (int t = 1; t < seq; ++t) if (p[t] > best) { best = p[t]; pos = t; }
    } else {
        for (int t = 0; t < seq; ++t) if (p[t] == eos_id) { pos = t; break; }
    }
    row_index[s] = s * seq + pos;
}

// ============================ attention ======================================
// Single-tile softmax attention for T <= 64 tokens, head_dim = 64 (E6).
// One wave per (image, head); a 256-thread workgroup covers 4 heads of one image.
//   S^T = K Q^T   : MFMA A = K rows (key on the register axis), B = Q (query on the lane axis)
//                   Q arrives pre-scaled by d_h^-0.5 (folded into Wq, bq on the host).
//   softmax over keys = over registers + the 4 lane groups (fp32, E6 "dtype=float32")
//   O^T = V^T P^T : P^T is already the B operand of the next MFMA under the k-slot order
//                   key(s,g,j) = 32 s + 16 (j>>2) + 4 g + (j&3); V^T fragments in that same
//                   order come from ds_read_b64_tr_b16 on a row-major [64 keys][64 d] LDS tile.
// qkv: [rows][3*hidden] bf16 (q | k | v);  out: [rows][hidden] bf16.
// q | k | v are read exactly once by the single-tile attention (59 MB per launch at batch 256): as streaming (nt) loads they do not
// displace the operand panels that the GEMMs of the other batches in flight are re-reading from the same L2s.  Build-time A/B switch.
#ifndef VQ_ATT_NT_LOAD
#define VQ_ATT_NT_LOAD 0
#endif
__device__ __forceinline__ uint4 ld16_att(const uint16_t* p) {
#if VQ_ATT_NT_LOAD
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return uint4{v[0], v[1], v[2], v[3]};
#else
    return *(const uint4*)p;
#endif
}

template <bool F16>
__global__ __launch_bounds__(256)          // (asked for 3 waves per SIMD it compiles to 117 registers without scratch - and runs the same 21.8 us)
void attention_t64_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                          int tokens, int hidden, int heads) {
    __shared__ __attribute__((aligned(16))) uint16_t vlds[4][64 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hgroups = heads >> 2;
    const int img = blockIdx.x / hgroups;
    const int head = (blockIdx.x - img * hgroups) * 4 + wave;
    const int ld = 3 * hidden;
    const uint16_t* base = qkv + (size_t)img * tokens * ld + head * 64;
    const int r16 = lane & 15, g = lane >> 4;
    typedef mfma_op<F16> op;
    typedef typename op::frag frag;

    // V tile -> LDS (rows >= tokens zero-filled: P is 0 there but 0*garbage may be NaN)
    uint16_t* vt = vlds[wave];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int id = it * 64 + lane, row = id >> 3, c = id & 7;
        uint4 val = {0u, 0u, 0u, 0u};
        if (row < tokens) val = ld16_att(base + 2 * hidden + (size_t)row * ld + c * 8);
        *(uint4*)(vt + row * 64 + c * 8) = val;
    }

    // K (A operand) and Q (B operand) fragments straight from global memory
    frag kf[4][2], qf[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = t * 16 + r16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 kv = {0u, 0u, 0u, 0u}, qv = {0u, 0u, 0u, 0u};
            if (row < tokens) {
                kv = ld16_att(base + hidden + (size_t)row * ld + ks * 32 + g * 8);
                qv = ld16_att(base + (size_t)row * ld + ks * 32 + g * 8);
            }
            kf[t][ks] = __builtin_bit_cast(frag, kv);
            qf[t][ks] = __builtin_bit_cast(frag, qv);
        }
    }

    // S^T[mt][nt]: lane holds keys 16 mt + 4 g + r (r = register), query 16 nt + r16
    f32x4 s[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = op::run(kf[mt][0], qf[nt][0], a);
            a = op::run(kf[mt][1], qf[nt][1], a);
            s[mt][nt] = a;
        }

    // softmax over keys, per query column
    float inv_sum[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float mx = -3.0e38f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = mt * 16 + g * 4 + r;
                if (key >= tokens) s[mt][nt][r] = -3.0e38f;
                mx = fmaxf(mx, s[mt][nt][r]);
            }
        mx = rows4_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[mt][nt][r] - mx);     // masked keys: exp(-huge) = 0
                s[mt][nt][r] = p;
                sum += p;
            }
        sum = rows4_sum(sum);
        inv_sum[nt] = 1.0f / sum;
    }

    // P^T fragments (B operand): k-slot j<4 -> tile 2s reg j ; j>=4 -> tile 2s+1 reg j-4
    frag pf[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const f32x4 lo = s[2 * ss][nt], hi = s[2 * ss + 1][nt];
            const uint2 plo = pack4_h<F16>(lo), phi = pack4_h<F16>(hi);
            pf[nt][ss] = __builtin_bit_cast(frag, uint4{plo.x, plo.y, phi.x, phi.y});
        }

    __syncthreads();   // V tile visible (all 64 lanes active from here on: tr reads need full EXEC)

    // V^T fragments (A operand) by transposed LDS reads.  Lane i=4q+p of a 16-lane group
    // addresses row q, columns 4p..4p+3 of a 4x16 block and receives column i of its 4 rows.
    const int q4 = r16 >> 2, p4 = r16 & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        f32x4 o[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int key0 = 32 * ss + 4 * g + q4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4*)(vt + (key0) * 64 + dt * 16 + p4 * 4));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4*)(vt + (key0 + 16) * 64 + dt * 16 + p4 * 4));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const frag vf = __builtin_bit_cast(frag, both);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                o[nt] = op::run(vf, pf[nt][ss], o[nt]);
        }
        // O^T: lane holds d = 16 dt + 4 g + r, query 16 nt + r16
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int qrow = nt * 16 + r16;
            if (qrow < tokens) {
                f32x4 v = o[nt];
                v[0] *= inv_sum[nt]; v[1] *= inv_sum[nt]; v[2] *= inv_sum[nt]; v[3] *= inv_sum[nt];
                *(uint2*)(out + ((size_t)img * tokens + qrow) * hidden + head * 64 + dt * 16 + g * 4) =
                    pack4_h<F16>(v);
            }
        }
    }
}

// ---- single-tile attention, T a compile-time constant: the instruction diet ----
// attention_t64_kernel above is not bound by memory but by instruction ISSUE: 9,216 (image, head) units on 1,024 SIMDs = 9 units
// per SIMD, and a unit was ~1,500 instructions (176 selects zeroing / masking padded rows, 128 accumulator-register reads because
// the MFMAs were compiled to their AGPR form, ~130 instructions of 64-bit address arithmetic, a sub + mul + exp per score, exec
// branches around 16 stores): 9 x ~5k cycles = the 22 us it ran in, whatever the loads did.  This form does the same arithmetic in
// ~1/3 of the instructions:  [r04]
//   * q | k | v through ONE buffer descriptor per wave whose range ends with the last token's V slice: rows past the sequence read
//     as zeros and stores of query rows past it are dropped by the range check — no clamps, selects or exec masks;
//   * V goes global -> LDS by buffer_load ... lds (7 instructions, no registers), the 16-byte chunks of a row XOR-swizzled by the
//     row pair so that the transposed fragment reads spread over all banks (row-major 128-byte rows were an 8-way conflict);
//   * T is a template parameter: scores of key positions no lane holds a live key for are never computed on (for T = 50 the
//     fourth key tile has 2 live registers of 16: 14 exponentials per query column instead of 16), p = exp2(fma(s, log2 e, -m log2 e));
//   * asked for two waves per SIMD the MFMAs compile to their VGPR form (no v_accvgpr_read);
//   * the V^T fragments are read so that a lane ends up with EIGHT consecutive head dimensions of a query row: 8 stores of 16 bytes per
//     wave instead of 16 of 8 (the store tail was issue-bound: 17.9 -> 15.7 us per launch, the largest single step after the diet).
// Same operand layouts as attention_t64_kernel (S^T = K Q^T, P^T the B operand of O^T = V^T P^T); no workgroup barrier: every LDS
// byte a wave reads was written by its own LDS-DMA.
// A operand of sixteen rows of ones: O^T gets a fifth d-tile whose every row is the column sum of P^T, i.e. the softmax
// denominator of the probabilities AS ROUNDED to 16 bits - from the matrix pipe, which has slack in every attention kernel here,
// instead of ~1.5 vector instructions per score (adds, the cross-row butterfly) on the pipe that bounds them.  [r04]
template <bool F16>
__device__ __forceinline__ typename mfma_op<F16>::frag ones_frag() {
    if constexpr (F16) { const _Float16 o = (_Float16)1.0f; return f16x8{o, o, o, o, o, o, o, o}; }
    else { const __bf16 o = (__bf16)1.0f; return bf16x8{o, o, o, o, o, o, o, o}; }
}
__device__ __forceinline__ uint4 buf_ld16(__amdgpu_buffer_rsrc_t r, int voff, int imm) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + imm, 0, 0);
    return uint4{v[0], v[1], v[2], v[3]};
}
template <bool F16, int T>
__global__ __launch_bounds__(256, 2)
void attention_tile_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int hidden, int heads) {
    static_assert(T >= 17 && T <= 64, "single key tile of 64");
    __shared__ __attribute__((aligned(1024))) uint16_t vlds[4][64 * 64];
    typedef mfma_op<F16> op;
    typedef typename op::frag frag;
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hgroups = heads >> 2;
    const int img = blockIdx.x / hgroups;
    const int head = (blockIdx.x - img * hgroups) * 4 + wave;
    const int ld = 3 * hidden;
    const int r16 = lane & 15, g = lane >> 4;
    // ranges: q | k | v of this image from this head's first column up to the end of the last token's V slice; the output likewise
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(qkv + (size_t)img * T * ld + head * 64), 0, ((T - 1) * ld + 2 * hidden + 64) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(out + (size_t)img * T * hidden + head * 64), 0, ((T - 1) * hidden + 64) * 2, 0x00020000);
    uint16_t* vt = vlds[wave];

    // K (A operand) and Q (B operand) fragments: lane (r16, g) holds row 16 t + r16, dims 32 ks + 8 g .. + 8
    const int row_bytes = ld * 2;
    const int kq_off = r16 * row_bytes + g * 16;
    uint4 kraw[4][2], qraw[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int o = kq_off + t * 16 * row_bytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kraw[t][ks] = buf_ld16(src, o + hidden * 2, ks * 64);
            qraw[t][ks] = buf_ld16(src, o, ks * 64);
        }
    }
    // V tile -> LDS, rows of 128 bytes; the lane that fills physical chunk c of row r fetches logical chunk c ^ (((r >> 1) & 3) << 1).
    // Rows from the last partly live 8-row piece on are zero-filled first (P is 0 there, but 0 * stale LDS may be NaN).
    constexpr int PIECES = (T + 7) / 8, ZROW0 = (T / 8) * 8;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ZROW0 < 64) {
#pragma unroll
        for (int i = 0; i < (64 - ZROW0) * 128 / 1024; ++i)
            *(uint4*)(vt + ZROW0 * 64 + i * 512 + lane * 8) = uint4{0u, 0u, 0u, 0u};
    }
    const int v_off = (lane >> 3) * row_bytes + hidden * 4 + (((lane & 7) ^ (((lane >> 4) & 3) << 1)) << 4);
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
        if (it * 8 >= ZROW0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero fill of these rows has landed
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (lds_void*)(vt + it * 512), 16, v_off + it * 8 * row_bytes, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the unit is in flight before the first MFMA waits for one (the scheduler had
                                         // moved the V pieces behind the first K / Q waits: a second trip to memory)

    frag kf[4][2], qf[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[t][ks] = __builtin_bit_cast(frag, kraw[t][ks]);
            qf[t][ks] = __builtin_bit_cast(frag, qraw[t][ks]);
        }
    // S^T[mt][nt]: lane holds keys 16 mt + 4 g + r (r = register), query 16 nt + r16.  Key tiles past the sequence are not computed.
    constexpr int KT = (T + 15) / 16;          // key tiles with a live key
    constexpr int FT = T / 16, R = T % 16;     // full tiles; live keys of the partial one
    f32x4 s[4][4];
#pragma unroll
    for (int mt = 0; mt < KT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = op::run(kf[mt][0], qf[nt][0], a);
            a = op::run(kf[mt][1], qf[nt][1], a);
            s[mt][nt] = a;
        }
    // softmax over keys, per query column, in the log2 domain
    const float NEGBIG = -3.0e38f, L2E = 1.44269504088896341f;
    float inv_sum[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        float mx = NEGBIG;
#pragma unroll
        for (int mt = 0; mt < KT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (mt == FT) {
                    if (r >= R) continue;                                   // no lane holds a live key in this register
                    if (4 * g + r >= R) s[mt][nt][r] = NEGBIG;
                }
                mx = fmaxf(mx, s[mt][nt][r]);
            }
        mx = rows4_max(mx);
        const float m2 = mx * L2E;
#pragma unroll
        for (int mt = 0; mt < KT; ++mt) {
            f32x4 e = s[mt][nt] * L2E - m2;
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = (mt == FT && r >= R) ? 0.f : __builtin_amdgcn_exp2f(e[r]);
            s[mt][nt] = e;
        }
    }
    // P^T fragments (B operand): k-slot j<4 -> tile 2s reg j ; j>=4 -> tile 2s+1 reg j-4
    frag pf[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const uint2 plo = 2 * ss < KT ? pack4_h<F16>(s[2 * ss][nt]) : uint2{0u, 0u};
            const uint2 phi = 2 * ss + 1 < KT ? pack4_h<F16>(s[2 * ss + 1][nt]) : uint2{0u, 0u};
            pf[nt][ss] = __builtin_bit_cast(frag, uint4{plo.x, plo.y, phi.x, phi.y});
        }
    // the denominators: column sums of P^T by MFMA against rows of ones (ones_frag), while the V tile is still landing
    {
        const frag ones = ones_frag<F16>();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
                if (2 * ss < KT) d = op::run(ones, pf[nt][ss], d);
            inv_sum[nt] = __builtin_amdgcn_rcpf(d[0]);
        }
    }

    __builtin_amdgcn_sched_barrier(0);                   // (nothing but arithmetic lies between the loads and this wait: unpinned, it floats up to them)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the V tile has landed (LDS-DMA counts in vmcnt)

    // V^T fragments (A operand) by transposed LDS reads.  Lane i = 4p + j of a 16-lane group: the lane (q, p) addresses row q and four
    // columns c0(p) .. c0(p) + 3, and lane i receives column j of lane p's four for the group's 4 rows - which d-column an A row means
    // is the reader's choice.  c0(p) = 32 a + 8 p + 4 b for the tile dt = 2 a + b: accumulator register r of lane (r16, g) then holds
    // d = 32 a + 8 g + 4 b + r, i.e. the two tiles of a pair give a lane EIGHT consecutive d of a query row - one 16-byte store per
    // (pair, query tile), 8 per wave instead of 16 of 8 bytes (the store tail of this shape is issue-bound: MI355X_MICROARCH.md).
    // Logical 16-byte chunk 4 a + p of a row sits at physical chunk (4 a + p) ^ (sw << 1), sw = ((row >> 1) & 3) =
    // (2 (g & 1) + (q >> 1)) & 3 for every row 32 ss + 16 h + 4 g + q this lane reads.
    const int q4 = r16 >> 2, p4 = r16 & 3;
    const int sw = (2 * (g & 1) + (q4 >> 1)) & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const int o_off = (r16 * hidden + g * 8) * 2;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        f32x4 o[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const uint16_t* vrow = vt + (4 * g + q4) * 64 + (((4 * a + p4) ^ (sw << 1)) << 3) + 4 * b;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) o[b][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                if (2 * ss >= KT) continue;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vrow + (32 * ss) * 64));
                s16x4 hi = {0, 0, 0, 0};
                if (2 * ss + 1 < KT) hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vrow + (32 * ss + 16) * 64));
                const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const frag vf = __builtin_bit_cast(frag, both);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) o[b][nt] = op::run(vf, pf[nt][ss], o[b][nt]);
            }
        }
        // O^T: lane holds d = 32 a + 8 g + 4 b + r, query 16 nt + r16; rows past the sequence fall outside dst's range
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt * 16 >= T) continue;
            const uint2 lo = pack4_h<F16>(o[0][nt] * inv_sum[nt]), hi = pack4_h<F16>(o[1][nt] * inv_sum[nt]);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, dst, o_off + nt * 16 * hidden * 2 + a * 64, 0, 0);
        }
    }
}

// ---- streaming attention: any T, head_dim 64, optional causal mask ----
// One wave per (image, head, 64-query tile); it walks the keys in tiles of 64 with an online softmax
// (running max m and sum l per query column, O rescaled when the max moves).  Same MFMA formulation as
// attention_t64_kernel: S^T = K Q^T with the query on the lane axis, P^T reused as the next MFMA's B
// operand, V^T fragments by ds_read_b64_tr_b16 from a per-wave row-major V tile in LDS.  No workgroup
// barrier: every LDS byte a wave reads was written by that wave (LDS executes a wave's accesses in order).
// Used for ViT-L/14@336 (577 tokens) and, with CAUSAL, for the CLIP text tower (77 tokens).
template <bool F16, bool CAUSAL>
__global__ __launch_bounds__(256)
void attention_stream_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                             int tokens, int hidden, int heads, int q_tiles, int total_units) {
    __shared__ __attribute__((aligned(16))) uint16_t vlds[4][64 * 64];
    typedef mfma_op<F16> op;
    typedef typename op::frag frag;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * 4 + wave;                 // ((img * heads) + head) * q_tiles + qt
    if (unit >= total_units) return;                        // wave-uniform; no workgroup barriers below
    const int qt = unit % q_tiles;
    const int ih = unit / q_tiles;
    const int head = ih % heads, img = ih / heads;
    const int ld = 3 * hidden;
    const uint16_t* base = qkv + (size_t)img * tokens * ld + head * 64;
    const int r16 = lane & 15, g = lane >> 4;
    const int q0 = qt * 64;
    uint16_t* vt = vlds[wave];

    frag qf[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = q0 + t * 16 + r16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 qv = {0u, 0u, 0u, 0u};
            if (row < tokens) qv = *(const uint4*)(base + (size_t)row * ld + ks * 32 + g * 8);
            qf[t][ks] = __builtin_bit_cast(frag, qv);
        }
    }
    f32x4 o[4][4];                                          // O^T[dt][nt]: d = 16 dt + 4 g + r, query 16 nt + r16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float NEGBIG = -3.0e38f;
    float m_run[4] = {NEGBIG, NEGBIG, NEGBIG, NEGBIG}, l_run[4] = {0.f, 0.f, 0.f, 0.f};

    const int k_tiles = CAUSAL ? qt + 1 : (tokens + 63) / 64;
    const int q4 = r16 >> 2, p4 = r16 & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;

    for (int kt = 0; kt < k_tiles; ++kt) {
        const int k0 = kt * 64;
        // V tile -> LDS (rows beyond the sequence zero-filled)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int id = it * 64 + lane, row = id >> 3, c = id & 7;
            uint4 val = {0u, 0u, 0u, 0u};
            if (k0 + row < tokens) val = *(const uint4*)(base + 2 * hidden + (size_t)(k0 + row) * ld + c * 8);
            *(uint4*)(vt + row * 64 + c * 8) = val;
        }
        frag kf[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = k0 + t * 16 + r16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 kv = {0u, 0u, 0u, 0u};
                if (row < tokens) kv = *(const uint4*)(base + hidden + (size_t)row * ld + ks * 32 + g * 8);
                kf[t][ks] = __builtin_bit_cast(frag, kv);
            }
        }
        f32x4 sc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                a = op::run(kf[mt][0], qf[nt][0], a);
                a = op::run(kf[mt][1], qf[nt][1], a);
                sc[mt][nt] = a;
            }
        float alpha[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int query = q0 + nt * 16 + r16;
            float mx = NEGBIG;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + mt * 16 + g * 4 + r;
                    if (key >= tokens || (CAUSAL && key > query)) sc[mt][nt][r] = NEGBIG;
                    mx = fmaxf(mx, sc[mt][nt][r]);
                }
            mx = rows4_max(mx);
            const float m_new = fmaxf(m_run[nt], mx);
            alpha[nt] = __expf(m_run[nt] - m_new);          // 0 on the first tile (m_run = -3e38)
            float sum = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // a masked score equals NEGBIG exactly; exp(NEGBIG - m_new) must be 0 even when m_new is NEGBIG too
                    const float p = sc[mt][nt][r] <= NEGBIG ? 0.f : __expf(sc[mt][nt][r] - m_new);
                    sc[mt][nt][r] = p;
                    sum += p;
                }
            sum = rows4_sum(sum);
            l_run[nt] = l_run[nt] * alpha[nt] + sum;
            m_run[nt] = m_new;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                o[dt][nt][0] *= alpha[nt]; o[dt][nt][1] *= alpha[nt]; o[dt][nt][2] *= alpha[nt]; o[dt][nt][3] *= alpha[nt];
            }
        frag pf[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const uint2 plo = pack4_h<F16>(sc[2 * ss][nt]), phi = pack4_h<F16>(sc[2 * ss + 1][nt]);
                pf[nt][ss] = __builtin_bit_cast(frag, uint4{plo.x, plo.y, phi.x, phi.y});
            }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int key0 = 32 * ss + 4 * g + q4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt + key0 * 64 + dt * 16 + p4 * 4));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt + (key0 + 16) * 64 + dt * 16 + p4 * 4));
                const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const frag vf = __builtin_bit_cast(frag, both);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) o[dt][nt] = op::run(vf, pf[nt][ss], o[dt][nt]);
            }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int qrow = q0 + nt * 16 + r16;
        const float inv = 1.0f / l_run[nt];
        if (qrow < tokens) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 v = o[dt][nt];
                v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
                *(uint2*)(out + ((size_t)img * tokens + qrow) * hidden + head * 64 + dt * 16 + g * 4) = pack4_h<F16>(v);
            }
        }
    }
}

// ---- streaming attention, workgroup-cooperative ----
// Same arithmetic as attention_stream_kernel, but a 256-thread workgroup owns 4 consecutive query
// tiles of ONE (image, head): the K and V tiles of a key step are loaded once by the whole workgroup
// into a double-buffered LDS image (K rows XOR-swizzled for conflict-free ds_read_b128 fragments, V
// row-major for the transposed reads) and shared by its 4 waves, and the next key step's loads are in
// flight while the current one is consumed.  (The per-wave version re-read K and V from global memory
// once per query tile and ran at ~280 TFLOP/s on ViT-L/14@336, 32 % of that model's step.)
// NQ = 16-row query blocks per wave: 4 (64 rows, the form above) or 2.  With 32 rows per wave the kernel needs half the
// accumulator, query and score registers (three waves per SIMD instead of two; compiled for four it spills 9 dwords and is no
// faster) and 577 tokens pad to 19 x 32 = 608 rows instead of 640; the K and V tiles are then shared by 128 query rows per
// workgroup instead of 256 (twice the LDS-DMA traffic, which is small beside the softmax).  ViT-L/14@336, 32 frames: attention
// 3.33 -> 2.83 ms per step (0.126 -> 0.148 of the MFMA peak on the 577^2 count), 2,292 -> 2,338 frames/s.  NQ = 2 is
// instantiated for the non-causal (image) tower only; $VQ_AMD_ATTN=q64 selects the 64-row form for A/B.
template <bool F16, bool CAUSAL, int NQ = 4>
__global__ __launch_bounds__(256, NQ == 2 ? 4 : 2)
void attention_stream_wg_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                int tokens, int hidden, int heads, int q_tiles, int q_groups) {
    __shared__ __attribute__((aligned(16))) uint16_t klds[2][64 * 64];
    __shared__ __attribute__((aligned(16))) uint16_t vlds[2][64 * 64];
    typedef mfma_op<F16> op;
    typedef typename op::frag frag;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qg = blockIdx.x % q_groups;
    const int ih = blockIdx.x / q_groups;
    const int head = ih % heads, img = ih / heads;
    const int qt = qg * 4 + wave;                           // this wave's query tile (may be past the end)
    const bool live = qt < q_tiles;
    const int ld = 3 * hidden;
    const uint16_t* base = qkv + (size_t)img * tokens * ld + head * 64;
    const int r16 = lane & 15, g = lane >> 4;
    static_assert(NQ == 4 || (NQ == 2 && !CAUSAL), "32-row query tiles: image tower only (the causal bounds assume 64-row tiles)");
    constexpr int QROWS = 16 * NQ;
    const int q0 = qt * QROWS;

    // Rows past `tokens` are never written by the tile loads below; zero both buffers once so that whatever a
    // masked key position holds is finite (its probability is 0, and 0 * finite = 0 in the P.V product).
    for (int i = tid; i < 2 * 64 * 64 / 8; i += 256) {
        ((uint4*)&klds[0][0])[i] = uint4{0u, 0u, 0u, 0u};
        ((uint4*)&vlds[0][0])[i] = uint4{0u, 0u, 0u, 0u};
    }

    frag qf[NQ][2];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const int row = q0 + t * 16 + r16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 qv = {0u, 0u, 0u, 0u};
            if (live && row < tokens) qv = *(const uint4*)(base + (size_t)row * ld + ks * 32 + g * 8);
            qf[t][ks] = __builtin_bit_cast(frag, qv);
        }
    }
    f32x4 o[4][NQ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NQ; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float NEGBIG = -3.0e38f;
    const float L2E = 1.44269504088896341f;
    // running maximum in the log2 domain (score * log2 e): p = exp2(score * log2e - m2) is one fma + one v_exp
    // the running denominators are a fifth d-tile of the O^T accumulators (ones_frag): rescaled with them, read at the end
    float m_run[NQ];
    f32x4 osum[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) { m_run[j] = NEGBIG; osum[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const frag ones = ones_frag<F16>();

    const int all_k_tiles = (tokens + 63) / 64;
    const int last_q_tile = min(q_tiles - 1, qg * 4 + 3);
    const int wg_k_tiles = CAUSAL ? min(all_k_tiles, last_q_tile + 1) : all_k_tiles;   // key steps the workgroup walks
    const int my_k_tiles = !live ? 0 : (CAUSAL ? qt + 1 : all_k_tiles);

    // Tile loads by LDS-DMA (no register round trip): a 64 x 64 tile is 512 16-byte positions; thread tid fills
    // positions tid and 256 + tid (row = pos >> 3, physical chunk = pos & 7).  K is stored XOR-swizzled for
    // conflict-free ds_read_b128 fragments, so the lane that owns physical chunk c fetches logical chunk
    // c ^ ((row >> 1) & 7); V is row-major for the transposed reads.
    auto fetch = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int pos = sl * 256 + tid;
            const int row = pos >> 3, cp = pos & 7;
            const int grow = kt * 64 + row;
            if (grow < tokens) {
                const uint16_t* src = base + (size_t)grow * ld;
                __builtin_amdgcn_global_load_lds((gbl_void*)(src + hidden + (cp ^ ((row >> 1) & 7)) * 8),
                                                 (lds_void*)(&klds[buf][sl * 2048 + wave * 512]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(src + 2 * hidden + cp * 8),
                                                 (lds_void*)(&vlds[buf][sl * 2048 + wave * 512]), 16, 0, 0);
            }
        }
    };
    const int q4 = r16 >> 2, p4 = r16 & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int kfx = (r16 >> 1) & 7;

    __syncthreads();                                        // zero fill done before the first tile lands
    fetch(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // One key step.  EDGE = the masking of keys past the sequence end (and, causal, past the query) is compiled in: as a run-time
    // flag it was if-converted into a compare and a select per score on EVERY tile.  The image tower masks only in its LAST key
    // step, which is peeled off the loop below; the causal (text) tower, two key steps long, masks in every step.
    auto key_step = [&](int kt, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int buf = kt & 1;
        if (kt + 1 < wg_k_tiles) fetch(kt + 1, buf ^ 1);    // the other buffer was last read in step kt-1 (barrier below it)
        if (kt < my_k_tiles) {
            const int k0 = kt * 64;
            const uint16_t* kt_l = klds[buf];
            const uint16_t* vt = vlds[buf];
            frag kf[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    kf[t][ks] = *(const frag*)(kt_l + (t * 16 + r16) * 64 + (((ks * 4 + g) ^ kfx) * 8));
            // the 64 query rows of the wave are independent: two passes of 32 rows keep the live score /
            // probability registers at half (the whole kernel then fits 2 waves per SIMD)
#pragma unroll
            for (int hh = 0; hh < NQ / 2; ++hh) {
                f32x4 sc[4][2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 a = {0.f, 0.f, 0.f, 0.f};
                        a = op::run(kf[mt][0], qf[2 * hh + j][0], a);
                        a = op::run(kf[mt][1], qf[2 * hh + j][1], a);
                        sc[mt][j] = a;
                    }
                float alpha[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nt = 2 * hh + j;
                    const int query = q0 + nt * 16 + r16;
                    float mx = NEGBIG;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (EDGE) {
                                const int key = k0 + mt * 16 + g * 4 + r;
                                if (key >= tokens || (CAUSAL && key > query)) sc[mt][j][r] = NEGBIG;
                            }
                            mx = fmaxf(mx, sc[mt][j][r]);
                        }
                    mx = rows4_max(mx);
                    const float m_new = fmaxf(m_run[nt], mx * L2E);
                    alpha[j] = __builtin_amdgcn_exp2f(m_run[nt] - m_new);
                    // p = exp2(s log2e - m); no running sum: osum  [r04]
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        f32x4 e;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float a = __builtin_fmaf(sc[mt][j][r], L2E, -m_new);
                            asm volatile("" : "+v"(a));          // plain v_fma_f32 (2.8 cycles): paired into v_pk_fma_f32 (4.9 for two) the step measured 1 % slower
                            float p = __builtin_amdgcn_exp2f(a);
                            if (EDGE && sc[mt][j][r] <= NEGBIG) p = 0.f;
                            e[r] = p;
                        }
                        sc[mt][j] = e;
                    }
                    m_run[nt] = m_new;
                }
                // the running maximum settles after the first tiles: rescale the accumulators only when some row's moved
                if (__builtin_amdgcn_ballot_w64(alpha[0] != 1.0f || alpha[1] != 1.0f) != 0) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4& oo = o[dt][2 * hh + j];
                            oo[0] *= alpha[j]; oo[1] *= alpha[j]; oo[2] *= alpha[j]; oo[3] *= alpha[j];
                        }
#pragma unroll
                    for (int j = 0; j < 2; ++j) osum[2 * hh + j][0] *= alpha[j];       // (its four registers hold the same sum: one is read)
                }
                frag pf[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss) {
                        const uint2 plo = pack4_h<F16>(sc[2 * ss][j]), phi = pack4_h<F16>(sc[2 * ss + 1][j]);
                        pf[j][ss] = __builtin_bit_cast(frag, uint4{plo.x, plo.y, phi.x, phi.y});
                        osum[2 * hh + j] = op::run(ones, pf[j][ss], osum[2 * hh + j]);
                    }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss) {
                        const int key0 = 32 * ss + 4 * g + q4;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt + key0 * 64 + dt * 16 + p4 * 4));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt + (key0 + 16) * 64 + dt * 16 + p4 * 4));
                        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const frag vf = __builtin_bit_cast(frag, both);
#pragma unroll
                        for (int j = 0; j < 2; ++j) o[dt][2 * hh + j] = op::run(vf, pf[j][ss], o[dt][2 * hh + j]);
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile has landed
        __syncthreads();
    };
    if constexpr (CAUSAL) {
        for (int kt = 0; kt < wg_k_tiles; ++kt) key_step(kt, std::true_type{});
    } else {
        for (int kt = 0; kt + 1 < wg_k_tiles; ++kt) key_step(kt, std::false_type{});
        key_step(wg_k_tiles - 1, std::true_type{});
    }
    if (!live) return;
#pragma unroll
    for (int nt = 0; nt < NQ; ++nt) {
        const int qrow = q0 + nt * 16 + r16;
        const float inv = 1.0f / osum[nt][0];
        if (qrow < tokens) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 v = o[dt][nt];
                v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
                *(uint2*)(out + ((size_t)img * tokens + qrow) * hidden + head * 64 + dt * 16 + g * 4) = pack4_h<F16>(v);
            }
        }
    }
}

// Generic patch extraction (any patch size, e.g. 14): one thread per (image, patch, channel, ky) writes
// the ps values of that patch row and, for ky == 0 of channel 0, zero-fills the K padding of the row.
template <bool F16>
__global__ __launch_bounds__(256)
void patchify_generic_kernel(const uint8_t* __restrict__ frames, uint16_t* __restrict__ out,
                             int n, int S, int ps, int k_pad, int swap_rb) {
    const int grid = S / ps;
    const int64_t total = (int64_t)n * grid * grid * 3 * ps;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int ky = (int)(t % ps);
        int64_t u = t / ps;
        const int c = (int)(u % 3); u /= 3;
        const int gx = (int)(u % grid); u /= grid;
        const int gy = (int)(u % grid);
        const int b = (int)(u / grid);
        const int cs = swap_rb ? 2 - c : c;
        const uint8_t* src = frames + (((int64_t)b * S + gy * ps + ky) * S + gx * ps) * 3 + cs;
        uint16_t* dst = out + ((size_t)(b * grid + gy) * grid + gx) * k_pad + (c * ps + ky) * ps;
        for (int kx = 0; kx < ps; ++kx) dst[kx] = to_h16<F16>((float)((int)src[kx * 3] - 128));
        if (c == 0 && ky == 0) {
            uint16_t* pad = out + ((size_t)(b * grid + gy) * grid + gx) * k_pad + 3 * ps * ps;
            for (int i = 0; i < k_pad - 3 * ps * ps; ++i) pad[i] = 0;
        }
    }
}

// ============================ pooling head ===================================
// CLS token -> post_layernorm -> visual_projection (fp32 weights, no bias) ->
// L2 normalise (x / max(||x||, 1e-12), F.normalize)                 (E8-E10)
// Two launches.  pool_project_kernel: workgroup (image group of 8, output chunk of 64); each wave
// LayerNorms two CLS rows into LDS, then thread (o = tid & 63, pair = tid >> 6) accumulates output
// chunk*64 + o for images 2*pair, 2*pair+1 over k; the projection is stored TRANSPOSED
// [hidden][proj_dim] so the 64 threads of a wave read 256 contiguous bytes per k.
// l2_normalize_rows_kernel: one wave per image.
constexpr int POOL_IMGS = 8;
constexpr int POOL_CHUNK = 64;
template <int NV>
__global__ __launch_bounds__(256)
void pool_project_kernel(const float* __restrict__ x, const float* __restrict__ g,
                         const float* __restrict__ b, const float* __restrict__ wproj_t,
                         float* __restrict__ feat /*[n][proj_dim], un-normalised*/,
                         int n_images, int tokens, int proj_dim, float eps,
                         const int* __restrict__ row_index = nullptr /*text tower: EOS row of every sequence*/) {
    constexpr int H = NV * 256;
    __shared__ __attribute__((aligned(16))) float xn[POOL_IMGS][H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img0 = blockIdx.x * POOL_IMGS;
    const int o = blockIdx.y * POOL_CHUNK + lane;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int im = wave * 2 + j;
        const int img = min(img0 + im, n_images - 1);            // tail group: duplicate the last image
        float4 v[NV];
        const size_t row = row_index ? (size_t)row_index[img] : (size_t)img * tokens;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *(const float4*)(x + row * H + (i * 64 + lane) * 4);
        ln_row<NV>(v, g, b, lane, eps, H);
#pragma unroll
        for (int i = 0; i < NV; ++i) *(float4*)(&xn[im][(i * 64 + lane) * 4]) = v[i];
    }
    __syncthreads();
    const bool live = o < proj_dim;
    const float* wcol = wproj_t + (live ? o : 0);
    const float* x0 = xn[wave * 2], *x1 = xn[wave * 2 + 1];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < H; k += 4) {
        const float w0 = wcol[(size_t)(k + 0) * proj_dim], w1 = wcol[(size_t)(k + 1) * proj_dim];
        const float w2 = wcol[(size_t)(k + 2) * proj_dim], w3 = wcol[(size_t)(k + 3) * proj_dim];
        const float4 u = *(const float4*)(x0 + k), v = *(const float4*)(x1 + k);      // LDS broadcasts
        a0 += u.x * w0; a0 += u.y * w1; a0 += u.z * w2; a0 += u.w * w3;
        a1 += v.x * w0; a1 += v.y * w1; a1 += v.z * w2; a1 += v.w * w3;
    }
    if (live) {
        const int i0 = img0 + wave * 2, i1 = i0 + 1;
        if (i0 < n_images) feat[(size_t)i0 * proj_dim + o] = a0;
        if (i1 < n_images) feat[(size_t)i1 * proj_dim + o] = a1;
    }
}

inline __global__ __launch_bounds__(256)
void l2_normalize_rows_kernel(float* __restrict__ feat, uint16_t* __restrict__ out_f16, int n_images, int proj_dim) {
    const int lane = threadIdx.x & 63;
    const int img = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (img >= n_images) return;
    float* row = feat + (size_t)img * proj_dim;
    float ss = 0.f;
    for (int i = lane; i < proj_dim; i += 64) ss += row[i] * row[i];
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int i = lane; i < proj_dim; i += 64) {
        const float e = row[i] / nrm;
        row[i] = e;
        if (out_f16) out_f16[(size_t)img * proj_dim + i] = __builtin_bit_cast(uint16_t, (_Float16)e);
    }
}

}  // namespace vq

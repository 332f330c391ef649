// Shared host/device helpers for libvq_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/vq_amd.h"
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>

namespace vq {

// ---- error plumbing -------------------------------------------------------
// Every C-ABI entry point returns 0 or a negative code and leaves a message in
// a thread-local buffer (vq_last_error), mirroring the reference convention
// "log + raise" (reference src/core/feature_extractor.py:175-177).
// Codes are the VQ_* macros of include/vq_amd.h (INVALID: bad argument or unsupported
// geometry; HIP: runtime failure; STATE: library not initialised; OOM).

std::string& last_error();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define VQ_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ::vq::fail(VQ_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                              hipGetErrorString(e__), __FILE__, __LINE__);                \
    } while (0)

#define VQ_CHECK(cond, ...)                                                               \
    do {                                                                                  \
        if (!(cond)) return ::vq::fail(VQ_ERR_INVALID, __VA_ARGS__);                \
    } while (0)

#define VQ_TRY(expr)                                                                      \
    do {                                                                                  \
        int rc__ = (expr);                                                                \
        if (rc__ != 0) return rc__;                                                       \
    } while (0)

// ---- device-side types ----------------------------------------------------
typedef uint16_t bf16_t;   // raw bf16 bits
typedef uint16_t f16_t;    // raw fp16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__host__ __device__ inline uint16_t f32_to_bf16_rne(float f) {
    // round-to-nearest-even on the fp32 bits; NaN stays NaN (quiet)
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(uint16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}

// ---- result ordering of the k-NN path (knn_kernels.h, knn_scan_f16.h, vq_comm.hip) ----
// (distance, row) -> one 64-bit key whose unsigned order is the lexicographic
// (distance asc, row asc) order of hnsw.py:269 `sorted(candidates)[:k]`.
__host__ __device__ inline uint64_t dist_key(float d, uint32_t row) {
    uint32_t u = __builtin_bit_cast(uint32_t, d);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // monotone float -> uint
    return ((uint64_t)u << 32) | row;
}
__host__ __device__ inline float key_dist(uint64_t k) {
    uint32_t u = (uint32_t)(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, u);
}

// Tie order.  The reference sorts (distance, id) tuples, so equal distances come back in the order of the CALLER's ids —
// strings f"{video_id}_{i}" under video_search_system.py:164-166, where "video0_10" < "video0_2".  The device knows rows
// only; vq_index_set_id_ranks hands it rank[row] = position of the row's id in the caller's id order (and the library keeps
// the inverse).  Keys then carry the rank where they used to carry the row, every selection stays what it was, and the
// kernels that emit ids translate back.  Both pointers null = ids ARE the row numbers.
struct TieOrder {
    const int32_t* rank;   // [n] rank of row r
    const int32_t* row;    // [n] row of rank t
};
__device__ __forceinline__ uint32_t tie_of(const TieOrder t, int64_t row) { return t.rank ? (uint32_t)t.rank[row] : (uint32_t)row; }
__device__ __forceinline__ int32_t tie_row(const TieOrder t, uint32_t tie) { return t.row ? t.row[tie] : (int32_t)tie; }
// (distance, id) order of two scored rows: does (dj, rj) come before (di, ri)?  The ranks are fetched only when the distances
// are bit-equal and the rows differ — duplicates of a frame; rank-counting loops pay nothing for them otherwise.
__device__ __forceinline__ bool scored_before(const TieOrder t, float dj, int rj, float di, int ri) {
    const uint32_t hj = (uint32_t)(dist_key(dj, 0) >> 32), hi = (uint32_t)(dist_key(di, 0) >> 32);
    if (hj != hi) return hj < hi;
    if (rj == ri) return false;
    return t.rank ? t.rank[rj] < t.rank[ri] : rj < ri;
}

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Sum over the 16 lanes of a DPP row, the total in every lane: v_add_f32 with a DPP-permuted source (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror) instead of four ds_bpermute_b32 round trips through the LDS crossbar
// (what __shfl_xor compiles to: a residual-GEMM workgroup issued 256 of them per wave and waited on each).  Every step adds
// the same operand pairs as the xor butterfly 1, 2, 4, 8, so the result is bit-identical to it.
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
    return x;
}

// ... over the 8 lanes of a half row (lanes 8j .. 8j + 7), the total in each of them: the first three steps of the above.
__device__ __forceinline__ float row8_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    return x;
}

// The largest x of a wave, in every lane: four DPP steps inside each row of 16, then the four row leaders by v_readlane
// (no LDS crossbar).  x must not be NaN (v_max would drop it).
__device__ __forceinline__ float wave64_max(float x) {
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)));
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)));
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true)));
    x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true)));
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16)),
                r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Reductions ACROSS the four 16-lane rows of a wave (lane ^ 16, then lane ^ 32), the result in every lane: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange odd and even rows (upper and lower halves) of two registers in one vector
// instruction; with both operands the same value, the two results hold {self, partner} in some order in every lane, so a
// commutative combine of them is the xor-butterfly step — same operand pairs as __shfl_xor(x, 16) / (x, 32), bit-identical,
// without the two ds_bpermute_b32 round trips through the LDS crossbar (~100+ cycles each on the softmax's critical path).
// Written as inline asm on two in/out registers: through __builtin_amdgcn_permlane{16,32}_swap this compiler (ROCm 7.2) folds
// op(result[0], result[1]) to op(result[0], result[0]) whenever both inputs hold the same value (scripts/ubench/rows4_check.hip
// showed max = own value and sum = 4 x own).  The s_nop pads the VALU-write -> permlane-read hazard the asm hides from the
// compiler's hazard recogniser.
__device__ __forceinline__ void rows_swap16(float x, float& a, float& b) {
    a = x; b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void rows_swap32(float x, float& a, float& b) {
    a = x; b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// (the maximum is taken inside the asm: fmaxf() on the asm's opaque outputs made the compiler canonicalise both of them first —
// v_max_f32 x, x, x twice per step, 4 of the 8 instructions of a rows4_max)
__device__ __forceinline__ float rows4_max(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
        "s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    return a;
}
__device__ __forceinline__ float rows4_sum(float x) {
    float a, b;
    rows_swap16(x, a, b); x = a + b;
    rows_swap32(x, a, b); return a + b;
}

}  // namespace vq

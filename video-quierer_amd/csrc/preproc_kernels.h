// Frame preprocessing upstream of the encoder: Pillow's separable 8-bit resample
// (src/libImaging/Resample.c) as two integer passes, and the frame-quality statistics of
// reference src/core/frame_extractor.py:301-316.  Byte/integer work, HBM-bound: a source frame is read once
// (staged through LDS row by row), the uint8 intermediate is the only extra traffic.
#pragma once
#include "vq_common.h"

namespace vq {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;       // Resample.c PRECISION_BITS: 22-bit fixed-point weights
constexpr int RS_THREADS = 256;

__device__ __forceinline__ uint8_t rs_clip8(int v) {
    v >>= RS_PRECISION_BITS;                          // arithmetic shift, as clip8() does
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Horizontal pass (ImagingResampleHorizontal_8bpc, 3 bands).
// A workgroup (4 waves) owns RSH_ROWS = 64 source rows x one segment of `oc` output columns.  It stages the
// source bytes those columns touch into LDS, one dword-aligned row per LDS row (row pitch an odd number of
// dwords), then LANE = ROW: a wave takes output columns of the segment in turn, so the tap window and the
// weights are wave-uniform (the weights sit in LDS, read as broadcasts) and every lane streams consecutive dwords of its own row,
// bank-conflict free.  Four taps = 12 bytes = three dwords, realigned with v_alignbyte by the row's byte
// offset; each byte costs one extract and one multiply-add.  Results go through an LDS tile so the global
// stores are row-contiguous.
//   src  [n][h][w][3]; the pass covers source rows [row_first, row_first + rows_needed) of every frame
//   tmp  [n][rows_needed][out_cols][3] for output columns [col_first, col_first + out_cols)
//   kk rows are `ksize` ints, ksize a multiple of 4, zero beyond the tap count
constexpr int RSH_ROWS = 64;
typedef __attribute__((address_space(3))) void rs_lds_t;
typedef const __attribute__((address_space(1))) void rs_gbl_t;

template <bool DWORD_STORE>
__global__ __launch_bounds__(RS_THREADS)
void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                       int h, int w, int row_first, int rows_needed, int col_first, int out_cols,
                       int oc, int pitch_dw, int tile_pitch) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
    uint8_t* tile = (uint8_t*)(lds32 + RSH_ROWS * pitch_dw);        // [64][tile_pitch] bytes
    int* kl = (int*)(tile + RSH_ROWS * tile_pitch);                  // [oc][ksize] weights of this segment's columns
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int img = blockIdx.z, r0 = blockIdx.y * RSH_ROWS;
    const int xo0 = blockIdx.x * oc, ncol = min(oc, out_cols - xo0);
    const int xx0 = col_first + xo0;
    const int sx0 = bounds[2 * xx0];                                 // first source pixel the segment touches
    const int sx1 = bounds[2 * (xx0 + ncol - 1)] + bounds[2 * (xx0 + ncol - 1) + 1];
    const int span_bytes = (sx1 - sx0) * 3;
    const size_t pitch = (size_t)w * 3;
    const uint8_t* seg0 = src + ((size_t)img * h + row_first) * pitch + (size_t)sx0 * 3;

    // ---- stage: wave w copies rows w, w+4, ... with LDS-DMA dword loads (64 lanes -> 64 consecutive LDS dwords,
    // no VGPR round trip).  A row starts at the aligned dword that holds its first byte; an aligned dword that
    // contains a valid byte never leaves that byte's page, so the up-to-3 bytes read before / after the span
    // are harmless. ----
    for (int rr = wave; rr < RSH_ROWS; rr += 4) {
        const int row = min(r0 + rr, rows_needed - 1);               // rows past the end repeat the last one, never stored
        const uint8_t* g = seg0 + (size_t)row * pitch;
        const int a = (int)((uintptr_t)g & 3);
        const uint8_t* ga = g - a;
        const int nd = (a + span_bytes + 3) >> 2;
        for (int j0 = 0; j0 < nd; j0 += 64)
            if (j0 + lane < nd)
                __builtin_amdgcn_global_load_lds((rs_gbl_t*)(ga + 4 * (size_t)(j0 + lane)),
                                                 (rs_lds_t*)(lds32 + rr * pitch_dw + j0), 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = threadIdx.x; i < ncol * ksize; i += RS_THREADS) kl[i] = kk[(size_t)xx0 * ksize + i];
    __syncthreads();

    // ---- compute: lane = row ----
    const int my_row = min(r0 + lane, rows_needed - 1);
    const int a = (int)((uintptr_t)(seg0 + (size_t)my_row * pitch) & 3);
    const uint32_t* rowp = lds32 + lane * pitch_dw;
    for (int c = wave; c < ncol; c += 4) {
        const int xx = xx0 + c;
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const int4* k = (const int4*)(kl + c * ksize);              // wave-uniform address: an LDS broadcast read
        const int off = a + (xmin - sx0) * 3;
        int di = off >> 2;
        const int sh = off & 3;
        int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        uint32_t d0 = rowp[di];
        const int nchunk = (cnt + 3) >> 2;
        for (int q = 0; q < nchunk; ++q) {
            const uint32_t d1 = rowp[di + 1], d2 = rowp[di + 2], d3 = rowp[di + 3];
            const uint32_t e0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
            const uint32_t e1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
            const uint32_t e2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
            const int4 kq = k[q];
            const int k0 = kq.x, k1 = kq.y, k2 = kq.z, k3 = kq.w;
            // weights are 23-bit signed, pixels 8-bit: v_mad_i32_i24 (full rate) instead of a 32-bit multiply
            s0 += __mul24((int)(e0 & 255), k0);         s1 += __mul24((int)((e0 >> 8) & 255), k0);  s2 += __mul24((int)((e0 >> 16) & 255), k0);
            s0 += __mul24((int)(e0 >> 24), k1);         s1 += __mul24((int)(e1 & 255), k1);         s2 += __mul24((int)((e1 >> 8) & 255), k1);
            s0 += __mul24((int)((e1 >> 16) & 255), k2); s1 += __mul24((int)(e1 >> 24), k2);         s2 += __mul24((int)(e2 & 255), k2);
            s0 += __mul24((int)((e2 >> 8) & 255), k3);  s1 += __mul24((int)((e2 >> 16) & 255), k3); s2 += __mul24((int)(e2 >> 24), k3);
            d0 = d3;
            di += 3;
        }
        uint8_t* o = tile + lane * tile_pitch + c * 3;
        o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
    }
    __syncthreads();

    // ---- store the 64 x ncol tile, row-contiguous ----
    const int rows = min(RSH_ROWS, rows_needed - r0);
    uint8_t* out0 = tmp + (((size_t)img * rows_needed + r0) * out_cols + xo0) * 3;
    const size_t out_pitch = (size_t)out_cols * 3;
    if constexpr (DWORD_STORE) {
        const int nd = (ncol * 3) >> 2;                              // <= 24 dwords: half a wave per row
        const int j = lane & 31;
        for (int r = wave * 2 + (lane >> 5); r < rows; r += 8)
            if (j < nd) *(uint32_t*)(out0 + r * out_pitch + 4 * j) = *(const uint32_t*)(tile + r * tile_pitch + 4 * j);
    } else {
        const int nb = ncol * 3;
        for (int r = wave; r < rows; r += 4)
            for (int j = lane; j < nb; j += 64) out0[r * out_pitch + j] = tile[r * tile_pitch + j];
    }
}

// Horizontal pass, row-block-major form.  Measured on resample_h_kernel with s_memtime stamps (1080p -> 224, 64 frames):
// of a workgroup's 23-26k cycles, 11-14k go to ISSUING its 48 LDS-DMA requests per wave — the requests queue behind a
// memory system that serves this pattern (a ~570-byte piece of each of 64 rows 5,760 bytes apart, per workgroup, with
// the pieces of one row spread over workgroups on different XCDs) at ~2 TB/s — and 6.8k to the multiply-adds.  Here a
// workgroup keeps its RSH_ROWS rows and walks `spw` consecutive column segments left to right: what it asks for next
// continues where its last request ended (the halo lines of a segment boundary are in its XCD's L2), a row's span is
// ONE wave-wide request of aligned 16-byte loads held in registers, and the next segment's pixels and weights are in
// flight while the current one is multiplied.  Same LDS image, lane = row compute loop and tile store as above.
template <bool DWORD_STORE>
__global__ __launch_bounds__(RS_THREADS)
void resample_hx_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                        const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                        int h, int w, int row_first, int rows_needed, int col_first, int out_cols,
                        int oc, int pitch_dw, int tile_pitch, int spw /* segments per workgroup */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
    uint8_t* tile = (uint8_t*)(lds32 + RSH_ROWS * pitch_dw);        // [64][tile_pitch] bytes
    int* kl = (int*)(tile + RSH_ROWS * tile_pitch);                  // [oc][ksize] weights of the current segment's columns
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int img = blockIdx.z, r0 = blockIdx.y * RSH_ROWS;
    const int nseg = (out_cols + oc - 1) / oc;
    const int s_begin = blockIdx.x * spw, s_end = min(nseg, s_begin + spw);
    const size_t pitch = (size_t)w * 3;
    const uint8_t* frame = src + ((size_t)img * h + row_first) * pitch;

    uint4 v[RSH_ROWS / 4];
    int kw[4];                                                       // host: oc * ksize <= 4 * RS_THREADS
    auto issue = [&](int sg) {                                       // segment sg: pixels of rows w, w+4, ... and its weights -> registers
        const int xo0 = sg * oc, ncol = min(oc, out_cols - xo0), xx0 = col_first + xo0;
        const int sx0 = bounds[2 * xx0];
        const int span_bytes = (bounds[2 * (xx0 + ncol - 1)] + bounds[2 * (xx0 + ncol - 1) + 1] - sx0) * 3;
#pragma unroll
        for (int i = 0; i < RSH_ROWS / 4; ++i) {
            const int row = min(r0 + wave + 4 * i, rows_needed - 1);     // rows past the end repeat the last one, never stored
            const uint8_t* g = frame + (size_t)row * pitch + (size_t)sx0 * 3;
            const int a = (int)((uintptr_t)g & 15);
            v[i] = uint4{0u, 0u, 0u, 0u};
            // an aligned 16-byte chunk that holds a valid byte never leaves that byte's page
            if (lane < ((a + span_bytes + 15) >> 4)) v[i] = *(const uint4*)(g - a + 16 * (size_t)lane);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = (int)threadIdx.x + RS_THREADS * u;
            kw[u] = idx < ncol * ksize ? kk[(size_t)xx0 * ksize + idx] : 0;
        }
    };
    if (s_begin < s_end) issue(s_begin);
    for (int sg = s_begin; sg < s_end; ++sg) {
        const int xo0 = sg * oc, ncol = min(oc, out_cols - xo0), xx0 = col_first + xo0;
        const int sx0 = bounds[2 * xx0];
        if (4 * lane < pitch_dw - 3) {
#pragma unroll
            for (int i = 0; i < RSH_ROWS / 4; ++i) {
                uint32_t* d = lds32 + (wave + 4 * i) * pitch_dw + 4 * lane;
                d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = (int)threadIdx.x + RS_THREADS * u;
            if (idx < oc * ksize) kl[idx] = kw[u];
        }
        __syncthreads();                                             // segment staged; every wave is done with the previous tile
        if (sg + 1 < s_end) issue(sg + 1);

        // ---- compute: lane = row ----
        const int my_row = min(r0 + lane, rows_needed - 1);
        const int a = (int)((uintptr_t)(frame + (size_t)my_row * pitch + (size_t)sx0 * 3) & 15);
        const uint32_t* rowp = lds32 + lane * pitch_dw;
        for (int c = wave; c < ncol; c += 4) {
            const int xx = xx0 + c;
            const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
            const int4* k = (const int4*)(kl + c * ksize);          // wave-uniform address: an LDS broadcast read
            const int off = a + (xmin - sx0) * 3;
            int di = off >> 2;
            const int sh = off & 3;
            int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
            uint32_t d0 = rowp[di];
            const int nchunk = (cnt + 3) >> 2;
            for (int q = 0; q < nchunk; ++q) {
                const uint32_t d1 = rowp[di + 1], d2 = rowp[di + 2], d3 = rowp[di + 3];
                const uint32_t e0 = __builtin_amdgcn_alignbyte(d1, d0, sh);
                const uint32_t e1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                const uint32_t e2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
                const int4 kq = k[q];
                const int k0 = kq.x, k1 = kq.y, k2 = kq.z, k3 = kq.w;
                s0 += __mul24((int)(e0 & 255), k0);         s1 += __mul24((int)((e0 >> 8) & 255), k0);  s2 += __mul24((int)((e0 >> 16) & 255), k0);
                s0 += __mul24((int)(e0 >> 24), k1);         s1 += __mul24((int)(e1 & 255), k1);         s2 += __mul24((int)((e1 >> 8) & 255), k1);
                s0 += __mul24((int)((e1 >> 16) & 255), k2); s1 += __mul24((int)(e1 >> 24), k2);         s2 += __mul24((int)(e2 & 255), k2);
                s0 += __mul24((int)((e2 >> 8) & 255), k3);  s1 += __mul24((int)((e2 >> 16) & 255), k3); s2 += __mul24((int)(e2 >> 24), k3);
                d0 = d3;
                di += 3;
            }
            uint8_t* o = tile + lane * tile_pitch + c * 3;
            o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
        }
        __syncthreads();

        // ---- store the 64 x ncol tile, row-contiguous ----
        const int rows = min(RSH_ROWS, rows_needed - r0);
        uint8_t* out0 = tmp + (((size_t)img * rows_needed + r0) * out_cols + xo0) * 3;
        const size_t out_pitch = (size_t)out_cols * 3;
        if constexpr (DWORD_STORE) {
            const int nd = (ncol * 3) >> 2;                          // <= 24 dwords: half a wave per row
            const int j = lane & 31;
            for (int r = wave * 2 + (lane >> 5); r < rows; r += 8)
                if (j < nd) *(uint32_t*)(out0 + r * out_pitch + 4 * j) = *(const uint32_t*)(tile + r * tile_pitch + 4 * j);
        } else {
            const int nb = ncol * 3;
            for (int r = wave; r < rows; r += 4)
                for (int j = lane; j < nb; j += 64) out0[r * out_pitch + j] = tile[r * tile_pitch + j];
        }
    }
}

// Vertical pass (ImagingResampleVertical_8bpc).  A thread produces VEC consecutive bytes of one output row: 16 when
// rows are 16-byte aligned (one wide load per tap, all taps independent: the pass is latency-bound on its ~11 taps),
// else 4 or 1.  tmp [n][rows_in][row_bytes], dst [n][out_rows][row_bytes] for output rows [row_first, row_first +
// out_rows); `row_shift` is the source row that tmp row 0 corresponds to.
// Each clipped byte goes through an opaque register move before packing: ROCm 7.2's backend otherwise folds
// clamp(x >> 22) pairs into v_ashr_pk_u8_i32, whose upper destination half it does not clear (bytes 2-3 of the
// packed dword came out as stale register contents on gfx950).
__device__ __forceinline__ uint32_t rs_pack4(int s0, int s1, int s2, int s3) {
    uint32_t b0 = rs_clip8(s0), b1 = rs_clip8(s1), b2 = rs_clip8(s2), b3 = rs_clip8(s3);
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

template <int VEC>
__global__ __launch_bounds__(RS_THREADS)
void resample_v_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ dst,
                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                       int rows_in, int row_bytes, int row_first, int out_rows, int row_shift) {
    const int img = blockIdx.z;
    int j, yo;
    if constexpr (VEC == 16) {                                      // threads cover (output row, 16-byte chunk) pairs
        const int per_row = row_bytes >> 4;
        const int idx = blockIdx.x * RS_THREADS + threadIdx.x;
        yo = idx / per_row;
        j = (idx - yo * per_row) << 4;
        if (yo >= out_rows) return;
    } else {
        j = (blockIdx.x * RS_THREADS + threadIdx.x) * VEC;
        yo = blockIdx.y;
        if (j >= row_bytes) return;
    }
    const int yy = row_first + yo;
    const int ymin = bounds[2 * yy] - row_shift, cnt = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    const uint8_t* p = tmp + ((size_t)img * rows_in + ymin) * row_bytes + j;
    uint8_t* o = dst + ((size_t)img * out_rows + yo) * row_bytes + j;
    if constexpr (VEC == 16) {
        int s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 1 << (RS_PRECISION_BITS - 1);
        for (int t = 0; t < cnt; ++t) {
            const uint4 v = *(const uint4*)(p + (size_t)t * row_bytes);
            const int c = k[t];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                s[4 * d] += __mul24((int)(w4[d] & 255), c);             s[4 * d + 1] += __mul24((int)((w4[d] >> 8) & 255), c);
                s[4 * d + 2] += __mul24((int)((w4[d] >> 16) & 255), c); s[4 * d + 3] += __mul24((int)(w4[d] >> 24), c);
            }
        }
        *(uint4*)o = uint4{rs_pack4(s[0], s[1], s[2], s[3]), rs_pack4(s[4], s[5], s[6], s[7]),
                           rs_pack4(s[8], s[9], s[10], s[11]), rs_pack4(s[12], s[13], s[14], s[15])};
    } else if constexpr (VEC == 4) {
        int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
        for (int t = 0; t < cnt; ++t) {
            const uint32_t v = *(const uint32_t*)(p + (size_t)t * row_bytes);
            const int c = k[t];
            s0 += __mul24((int)(v & 255), c); s1 += __mul24((int)((v >> 8) & 255), c); s2 += __mul24((int)((v >> 16) & 255), c); s3 += __mul24((int)(v >> 24), c);
        }
        *(uint32_t*)o = rs_pack4(s0, s1, s2, s3);
    } else {
        int s = 1 << (RS_PRECISION_BITS - 1);
        for (int t = 0; t < cnt; ++t) s += __mul24((int)p[(size_t)t * row_bytes], k[t]);
        *o = rs_clip8(s);
    }
}

// ---- cv2.resize(frame, (w, h)), INTER_LINEAR, 8-bit (reference frame_extractor.py:283-284) --------------
// OpenCV's two-tap fixed-point bilinear: 11-bit weights, 32-bit horizontal sums, the VResizeLinear<uchar>
// rounding.  No antialiasing, so a thread reads 4 source pixels per output pixel: one thread per output byte.
//   xofs/yofs: first tap per output column / row; wx, wy: the two short weights per column / row
//   out [n][crop_h][crop_w][3] = window (crop_top, crop_left) of the resized frame
__global__ __launch_bounds__(RS_THREADS)
void cv_resize_linear_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                             const int* __restrict__ xofs, const int* __restrict__ wx,
                             const int* __restrict__ yofs, const int* __restrict__ wy,
                             int h, int w, int crop_top, int crop_left, int crop_h, int crop_w) {
    const int j = blockIdx.x * RS_THREADS + threadIdx.x;            // byte inside the output row
    const int yo = blockIdx.y, img = blockIdx.z;
    if (j >= crop_w * 3) return;
    const int xo = j / 3, c = j - xo * 3;
    const int dx = crop_left + xo, dy = crop_top + yo;
    const int sx = xofs[dx], sx1 = min(sx + 1, w - 1);
    const int a0 = wx[2 * dx], a1 = wx[2 * dx + 1];
    const int sy = yofs[dy];
    const int y0 = min(max(sy, 0), h - 1), y1 = min(max(sy + 1, 0), h - 1);
    const int b0 = wy[2 * dy], b1 = wy[2 * dy + 1];
    const uint8_t* f = src + (size_t)img * h * w * 3;
    const uint8_t* r0 = f + (size_t)y0 * w * 3;
    const uint8_t* r1 = f + (size_t)y1 * w * 3;
    const int d0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
    const int d1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
    const int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
    dst[((size_t)img * crop_h + yo) * crop_w * 3 + j] = (uint8_t)v;
}

// exact 2x2 down-scale: cv2 routes INTER_LINEAR to INTER_AREA's fast path, (a + b + c + d + 2) >> 2
__global__ __launch_bounds__(RS_THREADS)
void cv_resize_half_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                           int h, int w, int crop_top, int crop_left, int crop_h, int crop_w) {
    const int j = blockIdx.x * RS_THREADS + threadIdx.x;
    const int yo = blockIdx.y, img = blockIdx.z;
    if (j >= crop_w * 3) return;
    const int xo = j / 3, c = j - xo * 3;
    const uint8_t* p = src + (((size_t)img * h + 2 * (crop_top + yo)) * w + 2 * (crop_left + xo)) * 3 + c;
    const size_t pitch = (size_t)w * 3;
    dst[((size_t)img * crop_h + yo) * crop_w * 3 + j] = (uint8_t)((p[0] + p[3] + p[pitch] + p[pitch + 3] + 2) >> 2);
}

// ---- frame quality (reference frame_extractor.py:301-316) ------------------------------------------------
// per frame: sum of all bytes (np.mean(frame)), and over the grey image g = BGR2GRAY(frame) the sums of
// L and L^2 where L = cv2.Laplacian(g, CV_64F) (aperture 1: the 4-neighbour stencil, BORDER_REFLECT_101).
// Grey conversion: OpenCV 4.x fixed point, (B*3735 + G*19235 + R*9798 + 2^14) >> 15.
__device__ __forceinline__ int gray_bgr(const uint8_t* p) {
    return (p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15;
}
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

// acc [n][3] int64: {sum bytes, sum L, sum L^2}; zeroed by the caller.  One thread per pixel.
__global__ __launch_bounds__(RS_THREADS)
void frame_quality_kernel(const uint8_t* __restrict__ frames, long long* __restrict__ acc, int h, int w) {
    const int img = blockIdx.y;
    const uint8_t* f = frames + (size_t)img * h * w * 3;
    long long sb = 0, sl = 0, sl2 = 0;
    const int total = h * w;
    for (int idx = blockIdx.x * RS_THREADS + threadIdx.x; idx < total; idx += gridDim.x * RS_THREADS) {
        const int y = idx / w, x = idx - y * w;
        const uint8_t* p = f + (size_t)idx * 3;
        sb += p[0] + p[1] + p[2];
        const int yu = reflect101(y - 1, h), yd = reflect101(y + 1, h);
        const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
        const int lap = gray_bgr(f + ((size_t)yu * w + x) * 3) + gray_bgr(f + ((size_t)yd * w + x) * 3) +
                        gray_bgr(f + ((size_t)y * w + xl) * 3) + gray_bgr(f + ((size_t)y * w + xr) * 3) - 4 * gray_bgr(p);
        sl += lap; sl2 += (long long)lap * lap;
    }
    // wave reduction, then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sb += __shfl_down(sb, off); sl += __shfl_down(sl, off); sl2 += __shfl_down(sl2, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&acc[img * 3 + 0], (unsigned long long)sb);
        atomicAdd((unsigned long long*)&acc[img * 3 + 1], (unsigned long long)sl);
        atomicAdd((unsigned long long*)&acc[img * 3 + 2], (unsigned long long)sl2);
    }
}

}  // namespace vq

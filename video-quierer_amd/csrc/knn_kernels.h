// Device kernels of the exact cosine k-NN scan.
//   distance = fp32(1 - fp32(dot(row, q))),  result = k smallest by (distance, row)
// dot is the fixed-order fp64 chain of oracle/knn_oracle.c, so ids and
// distances are bit-identical to the oracle.
#pragma once
#include "vq_common.h"

namespace vq {

// ---- row normalisation (HNSWIndex.add, hnsw.py:157) for device-resident rows ----
// n2 = fp64 chain of x*x in index order; x <- fp32(x / fp32(sqrt(n2))).
// One thread per row keeps the chain order; rows are staged through LDS in 64-float column panels (16-byte loads, a
// quarter row segment per lane) so the global reads stay coalesced.  The division pass is wave = row, 16 bytes per lane:
// round 2's form indexed it by a flat element number and paid a 64-bit integer division per element (0.96 ms per
// 250k x 512 block = 1.07 TB/s).
constexpr int NORM_ROWS = 128;   // rows (= threads) per workgroup
__global__ __launch_bounds__(NORM_ROWS)
void normalize_rows_kernel(float* __restrict__ rows, int64_t n, int dim) {
    __shared__ float tile[NORM_ROWS][65];
    __shared__ float nrm[NORM_ROWS];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * NORM_ROWS;
    const bool vec = (dim & 3) == 0 && (((uintptr_t)rows) & 15) == 0;
    double acc = 0.0;
    for (int d0 = 0; d0 < dim; d0 += 64) {
        if (vec) {                                            // 16 lanes x 16 bytes = one 64-float row segment
            for (int i = tid; i < NORM_ROWS * 16; i += NORM_ROWS) {
                const int rr = i >> 4, c4 = (i & 15) * 4;
                const int64_t r = r0 + rr;
                float4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < n && d0 + c4 < dim) v = *(const float4*)(rows + r * dim + d0 + c4);
                tile[rr][c4] = v.x; tile[rr][c4 + 1] = v.y; tile[rr][c4 + 2] = v.z; tile[rr][c4 + 3] = v.w;
            }
        } else {
            for (int i = tid; i < NORM_ROWS * 64; i += NORM_ROWS) {
                const int rr = i >> 6, cc = i & 63;
                const int64_t r = r0 + rr;
                tile[rr][cc] = (r < n && d0 + cc < dim) ? rows[r * dim + d0 + cc] : 0.f;
            }
        }
        __syncthreads();
        const int lim = min(64, dim - d0);
        for (int c = 0; c < lim; ++c) { const double v = (double)tile[tid][c]; acc += v * v; }
        __syncthreads();
    }
    nrm[tid] = (float)sqrt(acc);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int rr = wave; rr < NORM_ROWS; rr += NORM_ROWS / 64) {
        const int64_t r = r0 + rr;
        if (r >= n) break;
        const float d = nrm[rr];
        float* row = rows + r * dim;
        if (vec) {
            for (int c = lane * 4; c < dim; c += 256) {
                float4 v = *(const float4*)(row + c);
                v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;      // IEEE fp32 division
                *(float4*)(row + c) = v;
            }
        } else {
            for (int c = lane; c < dim; c += 64) row[c] = row[c] / d;
        }
    }
}

// ---- |row|^2 range of rows added WITHOUT normalisation (vq_index_add(normalize=0), HNSWIndex.load) ----
// The fp16 scan's error bound is stated for near-unit rows (knn_scan_f16.h scan_eps_unit): the index tracks the
// smallest and largest |row|^2 it holds and leaves the fp16 path when they stray.  One wave per row; the positive
// fp32 bit patterns order like the values, so atomicMin/atomicMax on the bits keep the range.
__global__ __launch_bounds__(256)
void row_norm_range_kernel(const float* __restrict__ rows, int64_t n, int dim, uint32_t* __restrict__ range /*[2]: min, max bits*/) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.f;
    for (int i = lane * 4; i < dim; i += 256) {
        const float4 v = *(const float4*)(rows + r * dim + i);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        if (!(s >= 0.f) || s > 3.0e38f) s = 3.0e38f;          // NaN / inf rows: out of any acceptable range
        const uint32_t b = __builtin_bit_cast(uint32_t, s);
        atomicMin(range, b);
        atomicMax(range + 1, b);
    }
}

// empty index on the device API: every slot is "no candidate" (hnsw.py:243-244 returns [])
__global__ void fill_no_result_kernel(int32_t* __restrict__ ids, float* __restrict__ dist, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { ids[i] = -1; dist[i] = __builtin_inff(); }
}

// fp32 master -> fp16 scan copy (round-to-nearest-even)
__global__ __launch_bounds__(256)
void rows_to_f16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t count4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = *(const float4*)(src + i * 4);
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        const h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *(uint2*)(dst + i * 4) = __builtin_bit_cast(uint2, h);
    }
}

// In-place replacement of stored rows (HNSWIndex.add of an id that is already there, hnsw.py:160 `self.data[node_id] =
// vector`): src[i] -> rows[row_no[i]] (fp32 master) and its fp16 scan copy, the same conversion as rows_to_f16_kernel, so an
// updated matrix is bit-identical to one built from scratch.  row_no holds distinct row numbers (the host drops all but the
// last update of a row).  One thread per float4.
__global__ __launch_bounds__(256)
void scatter_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ row_no, int64_t n, int dim,
                         float* __restrict__ rows, uint16_t* __restrict__ rows16) {
    const int per_row = dim >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * per_row; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row) * 4;
        const float4 v = *(const float4*)(src + r * dim + c);
        const int64_t dst = row_no[r] * dim + c;
        *(float4*)(rows + dst) = v;
        typedef __attribute__((ext_vector_type(4))) _Float16 h4;
        const h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *(uint2*)(rows16 + dst) = __builtin_bit_cast(uint2, h);
    }
}

// ---- exact distances ----
// Workgroup tile: 64 rows x 32 queries, 256 threads; thread (r = tid&63, g = tid>>6)
// owns row r and queries 8g..8g+7 (a wave shares g, so query reads are LDS broadcasts).
// The k loop runs in index order over 64-wide panels: acc += (double)x[i]*(double)q[i].
__global__ __launch_bounds__(256)
void exact_dist_kernel(const float* __restrict__ rows, int64_t n, int dim,
                       const float* __restrict__ queries, int nq,
                       float* __restrict__ dist /*[nq][ld]*/, int64_t ld) {
    __shared__ float xs[64][65];
    __shared__ double qs[32][64];
    const int tid = threadIdx.x, r = tid & 63, g = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int q0 = blockIdx.y * 32;
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    for (int d0 = 0; d0 < dim; d0 += 64) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int rr = i >> 6, cc = i & 63;
            const int64_t gr = row0 + rr;
            xs[rr][cc] = (gr < n && d0 + cc < dim) ? rows[gr * dim + d0 + cc] : 0.f;
        }
        for (int i = tid; i < 32 * 64; i += 256) {
            const int qq = i >> 6, cc = i & 63;
            qs[qq][cc] = (q0 + qq < nq && d0 + cc < dim) ? (double)queries[(int64_t)(q0 + qq) * dim + d0 + cc] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            const double xv = (double)xs[r][c];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += xv * qs[g * 8 + j][c];   // product exact in fp64
        }
        __syncthreads();
    }
    const int64_t gr = row0 + r;
    if (gr < n) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = q0 + g * 8 + j;
            if (q < nq) dist[(int64_t)q * ld + gr] = 1.0f - (float)acc[j];
        }
    }
}

// ---- exact selection ----
// Two levels so that a large matrix is not scanned k times by one workgroup per query:
//   select_chunk_kernel: workgroup (chunk, query) holds its 8192 distances in registers
//     as (distance, tie) keys — tie = the row, or the rank of the row's id (TieOrder, vq_common.h) — and extracts the chunk's k smallest in k rounds of
//     "smallest key strictly above the previous one" (keys are unique: row is in the key);
//   merge_topk_kernel: one workgroup per query does the same over the chunks' lists.
constexpr int SEL_CHUNK = 8192;
constexpr int SEL_PER_THREAD = SEL_CHUNK / 256;

__device__ __forceinline__ uint64_t block_min_u64(uint64_t v, uint64_t* red /*[4]*/, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t other = __shfl_xor(v, o);
        v = other < v ? other : v;
    }
    __syncthreads();                      // previous round's readers are done with red[]
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    uint64_t b = red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) b = red[w] < b ? red[w] : b;
    return b;
}

// ---- exact distances for a HANDFUL of queries (the reference caller's one `index.search(query, k*2)` at a time,
// video_search_system.py:297, on an index of a few thousand frames) ----
// exact_dist_kernel above is shaped for batches (32 queries per tile, scalar loads, a load -> barrier -> multiply -> barrier
// round trip per 64-dim panel): 49 us for ONE query over 4,000 rows, nearly all of it exposed memory latency.  Here: 64 rows x
// up to 8 queries per workgroup, the row panel fetched as 16-byte loads one panel AHEAD of the multiply (registers), the
// queries converted to fp64 once.  Thread (r = tid & 63, g = tid >> 6) owns row r and queries 2g, 2g + 1; the chain is the
// same index-order fp64 sum (one rounding to fp32 at the end), so the distances are the same bits.
constexpr int EDS_MAX_Q = 8;
__global__ __launch_bounds__(256)
void exact_dist_small_kernel(const float* __restrict__ rows, int64_t n, int dim,
                             const float* __restrict__ queries, int nq,
                             float* __restrict__ dist /*[nq][ld]*/, int64_t ld) {
    __shared__ float xs[2][64][65];
    extern __shared__ __attribute__((aligned(16))) double qd[];        // [EDS_MAX_Q][dim]
    const int tid = threadIdx.x, r = tid & 63, g = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int panels = (dim + 63) >> 6;
    // a thread fetches four 16-byte pieces of a 64 x 64 panel: piece p = tid + 256 u -> row p >> 4, floats (p & 15) * 4 ..+3
    float4 nx[4];
    auto fetch = [&](int pi) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = tid + 256 * u, rr = p >> 4, c4 = (p & 15) * 4;
            const int64_t gr = row0 + rr;
            const int c = pi * 64 + c4;
            nx[u] = float4{0.f, 0.f, 0.f, 0.f};
            if (gr < n && c < dim) nx[u] = *(const float4*)(rows + gr * dim + c);        // dim % 4 == 0 (vq_index_create)
        }
    };
    fetch(0);
    for (int i = tid; i < EDS_MAX_Q * dim; i += 256) {
        const int q = i / dim;
        qd[i] = q < nq ? (double)queries[i] : 0.0;
    }
    double acc0 = 0.0, acc1 = 0.0;
    const bool live0 = 2 * g < nq, live1 = 2 * g + 1 < nq;
    for (int pi = 0; pi < panels; ++pi) {
        float (*x)[65] = xs[pi & 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = tid + 256 * u, rr = p >> 4, c4 = (p & 15) * 4;
            x[rr][c4] = nx[u].x; x[rr][c4 + 1] = nx[u].y; x[rr][c4 + 2] = nx[u].z; x[rr][c4 + 3] = nx[u].w;
        }
        __syncthreads();                               // (two buffers: the next panel's stores cannot overtake this panel's reads)
        if (pi + 1 < panels) fetch(pi + 1);
        if (live0) {                                   // wave-uniform: a wave shares g
            const double* q0 = qd + (size_t)(2 * g) * dim + pi * 64;
            const double* q1 = q0 + dim;
            const int lim = min(64, dim - pi * 64);
            if (live1) {
                for (int c = 0; c < lim; ++c) { const double xv = (double)x[r][c]; acc0 += xv * q0[c]; acc1 += xv * q1[c]; }
            } else {
                for (int c = 0; c < lim; ++c) acc0 += (double)x[r][c] * q0[c];
            }
        }
    }
    const int64_t gr = row0 + r;
    if (gr < n) {
        if (live0) dist[(int64_t)(2 * g) * ld + gr] = 1.0f - (float)acc0;
        if (live1) dist[(int64_t)(2 * g + 1) * ld + gr] = 1.0f - (float)acc1;
    }
}

// ---- exact selection for the same case: ONE workgroup per query finds the k smallest (distance, tie) keys of n distances ----
// select_chunk_kernel + merge_topk_kernel extract the minimum k times (two barriers per round: 26 us at k = 10, 43 us at
// k = 20, whatever n).  Here: (1) every thread's smallest key; (2) T = the k-th smallest of the 256 thread minima — at least
// k keys are <= T, so the k smallest overall are; (3) the keys <= T (k of them plus a few) are collected and (4) ranked by
// counting.  More collected keys than the list holds (a pathological layout) -> the workgroup falls back to extracting
// minima from global memory.  n <= SEL_SMALL_MAX_N.
constexpr int SEL_SMALL_MAX_N = 32768;
constexpr int SEL_SMALL_LIST = 1024;
__global__ __launch_bounds__(256)
void select_small_kernel(const float* __restrict__ dist, int64_t ld, int64_t n, int k,
                         int32_t* __restrict__ ids, float* __restrict__ out_dist, const TieOrder tie) {
    __shared__ uint64_t mins[256];
    __shared__ uint64_t list[SEL_SMALL_LIST];
    __shared__ uint64_t red[4];
    __shared__ uint64_t thr_s;
    __shared__ int cnt_s;
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* d = dist + (int64_t)q * ld;
    int32_t* oi = ids + (int64_t)q * k;
    float* od = out_dist + (int64_t)q * k;
    // Both passes walk the distances as 16-byte pieces, four pieces per thread in flight (a one-element-per-iteration loop paid a
    // dependent L2 round trip per element: 17 us at 4,000 rows, 35 us at 16,000).  Piece p holds rows 4p .. 4p + 3.
    const int64_t pieces = (n + 3) >> 2;
    const bool vec = (ld & 3) == 0;                                   // rows of `dist` 16-byte aligned (ld is a multiple of 64)
    auto keys_of = [&](int64_t p, uint64_t (&key)[4]) __attribute__((always_inline)) {
        const int64_t r0 = p * 4;
        float v[4]; uint32_t t[4];
        if (vec && r0 + 3 < n) {
            const float4 f = *(const float4*)(d + r0);
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            if (tie.rank) { const int4 tt = *(const int4*)(tie.rank + r0); t[0] = tt.x; t[1] = tt.y; t[2] = tt.z; t[3] = tt.w; }
            else { t[0] = (uint32_t)r0; t[1] = (uint32_t)r0 + 1; t[2] = (uint32_t)r0 + 2; t[3] = (uint32_t)r0 + 3; }
#pragma unroll
            for (int e = 0; e < 4; ++e) key[e] = dist_key(v[e], t[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) key[e] = (p < pieces && r0 + e < n) ? dist_key(d[r0 + e], tie_of(tie, r0 + e)) : ~0ull;
        }
    };
    uint64_t mine = ~0ull;
    for (int64_t p0 = tid; p0 < pieces; p0 += 4 * 256) {
        uint64_t key[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = p0 + 256 * u;
            if (p < pieces) keys_of(p, key[u]);
            else { key[u][0] = key[u][1] = key[u][2] = key[u][3] = ~0ull; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) mine = key[u][e] < mine ? key[u][e] : mine;
    }
    mins[tid] = mine;
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    const int kk = (int)min((int64_t)k, n);                      // results that exist
    {
        int rank = 0;
        for (int j = 0; j < 256; ++j) { const uint64_t o = mins[j]; rank += (o < mine) || (o == mine && j < tid); }
        if (rank == min(kk, 256) - 1) thr_s = mine;              // ranks are a permutation: one writer
    }
    __syncthreads();
    const uint64_t thr = kk > 256 ? ~0ull - 1 : thr_s;           // k beyond the thread count: every real key competes
    for (int64_t p0 = tid; p0 < pieces; p0 += 4 * 256) {
        uint64_t key[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = p0 + 256 * u;
            if (p < pieces) keys_of(p, key[u]);
            else { key[u][0] = key[u][1] = key[u][2] = key[u][3] = ~0ull; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (key[u][e] <= thr) {
                    const int pos = atomicAdd(&cnt_s, 1);
                    if (pos < SEL_SMALL_LIST) list[pos] = key[u][e];
                }
    }
    __syncthreads();
    const int c = cnt_s;
    for (int j = kk + tid; j < k; j += 256) { oi[j] = -1; od[j] = __builtin_inff(); }
    if (c <= SEL_SMALL_LIST) {
        for (int i = tid; i < c; i += 256) {
            const uint64_t ki = list[i];
            int rank = 0;
            for (int j = 0; j < c; ++j) rank += list[j] < ki;    // keys are distinct (the tie word is a permutation of the rows)
            if (rank < kk) { oi[rank] = tie_row(tie, (uint32_t)ki); od[rank] = key_dist(ki); }
        }
        return;
    }
    uint64_t prev = 0;                                           // the slow way, from global memory: kk rounds of "smallest key above the previous"
    for (int j = 0; j < kk; ++j) {
        uint64_t best = ~0ull;
        for (int64_t r = tid; r < n; r += 256) {
            const uint64_t key = dist_key(d[r], tie_of(tie, r));
            if ((j == 0 || key > prev) && key < best) best = key;
        }
        best = block_min_u64(best, red, tid);
        if (tid == 0) { oi[j] = tie_row(tie, (uint32_t)best); od[j] = key_dist(best); }
        prev = best;
    }
}

__global__ __launch_bounds__(256)
void select_chunk_kernel(const float* __restrict__ dist, int64_t ld, int64_t n, int k, int nchunks,
                         uint64_t* __restrict__ partial /*[q][nchunks][k]*/, const TieOrder tie) {
    __shared__ uint64_t red[4];
    const int q = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x;
    const float* d = dist + (int64_t)q * ld;
    const int64_t base = (int64_t)chunk * SEL_CHUNK;
    uint64_t keys[SEL_PER_THREAD];
#pragma unroll
    for (int i = 0; i < SEL_PER_THREAD; ++i) {
        const int64_t r = base + i * 256 + tid;
        keys[i] = r < n ? dist_key(d[r], tie_of(tie, r)) : ~0ull;
    }
    uint64_t* out = partial + ((int64_t)q * nchunks + chunk) * k;
    uint64_t prev = 0;
    for (int j = 0; j < k; ++j) {
        uint64_t best = ~0ull;
#pragma unroll
        for (int i = 0; i < SEL_PER_THREAD; ++i)
            if ((j == 0 || keys[i] > prev) && keys[i] < best) best = keys[i];
        best = block_min_u64(best, red, tid);
        if (tid == 0) out[j] = best;
        prev = best;
        if (best == ~0ull) {                    // chunk exhausted (block-uniform)
            for (int jj = j + 1 + tid; jj < k; jj += 256) out[jj] = ~0ull;
            break;
        }
    }
}

__global__ __launch_bounds__(256)
void merge_topk_kernel(const uint64_t* __restrict__ partial, int nchunks, int k,
                       int32_t* __restrict__ ids, float* __restrict__ out_dist, const TieOrder tie) {
    __shared__ uint64_t red[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const uint64_t* p = partial + (int64_t)q * nchunks * k;
    const int total = nchunks * k;
    uint64_t prev = 0;
    for (int j = 0; j < k; ++j) {
        uint64_t best = ~0ull;
        for (int i = tid; i < total; i += 256) {
            const uint64_t key = p[i];
            if ((j == 0 || key > prev) && key < best) best = key;
        }
        best = block_min_u64(best, red, tid);
        if (tid == 0) {
            const int64_t o = (int64_t)q * k + j;
            if (best == ~0ull) { ids[o] = -1; out_dist[o] = __builtin_inff(); }
            else { ids[o] = tie_row(tie, (uint32_t)best); out_dist[o] = key_dist(best); }
        }
        prev = best;
        if (best == ~0ull) {
            for (int jj = j + 1 + tid; jj < k; jj += 256) { ids[(int64_t)q * k + jj] = -1; out_dist[(int64_t)q * k + jj] = __builtin_inff(); }
            break;
        }
    }
}

}  // namespace vq

// Device-side fallback of the fp16 search: the queries whose exactness proof did not close
// (rescore_verify outcome 2) are redone by an exact scan WITHOUT the host looking at the flags, so
// vq_index_search_device stays asynchronous in every mode.
//
//   collect_flags_kernel     flags[nq] -> ordered list of flagged query numbers + count + outcome counters
//   exact_fallback_kernel    grid (row splits, slot lanes): a workgroup scans its row range for 8 flagged queries at a
//                            time (thread = row, fixed-order fp64 chain: the arithmetic of exact_dist_kernel and of
//                            oracle/knn_oracle.c) and keeps each query's k best (distance, row) keys in LDS
//   fallback_merge_kernel    per flagged query: the k best keys of its row splits' lists -> ids / distances, written
//                            over the fast path's answer for that query
//
// Every workgroup reads the flagged count first and leaves at once when it is zero (the normal case: three short
// launches instead of a device-to-host copy and a stream synchronisation).
#pragma once
#include "vq_common.h"
#include "knn_kernels.h"

namespace vq {

constexpr int FB_QG = 8;          // flagged queries per row pass
constexpr int FB_TILE = 256;      // rows per tile = threads per workgroup
constexpr int FB_PANEL = 32;      // dims per LDS panel
constexpr int FB_KMAX = 100;      // largest k of the fp16 path (RV_K_MAX); with the row tile and the candidate lists this fills the 64 KiB of static LDS
constexpr int FB_SPLIT_ROWS = 2048;   // rows per workgroup of the bulk rounds
constexpr int FB_FAST_ROWS = 512;     // ... of the first round (the first FB_FAST_SLOTS flagged queries: the normal case is a handful)
constexpr int FB_FAST_SLOTS = 64;
constexpr int FB_MAX_SPLITS = 4096;
constexpr int FB_SLOT_LANES = 2;
// device counters: [0] flagged queries, [1..3] outcome counts (proven, proven after rescans, exact fallback)
constexpr int FB_NCOUNTERS = 4;

__global__ __launch_bounds__(1024)
void collect_flags_kernel(const int32_t* __restrict__ flags, int nq, int32_t* __restrict__ slots,
                          int32_t* __restrict__ counters) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    __shared__ int st[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    if (tid < 3) st[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nq; i0 += 1024) {
        const int i = i0 + tid;
        int f = i < nq ? flags[i] : 0;
        if (f < 0 || f > 2) f = 2;
        const bool flagged = i < nq && f == 2;
        if (i < nq) atomicAdd(&st[f], 1);
        const unsigned long long mask = __ballot(flagged);
        const int prefix = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __builtin_popcountll(mask);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (flagged) slots[off + prefix] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base_s += t; }
        __syncthreads();
    }
    if (tid == 0) { counters[0] = base_s; counters[1] = st[0]; counters[2] = st[1]; counters[3] = st[2]; }
}

__global__ __launch_bounds__(FB_TILE)
void exact_fallback_kernel(const float* __restrict__ rows, int64_t n, int dim,
                           const float* __restrict__ queries, const int32_t* __restrict__ slots,
                           const int32_t* __restrict__ counters, int slot_base, int slot_cap, int k,
                           int64_t rows_per_split, uint64_t* __restrict__ partial /*[slot_cap][splits][k]*/, const TieOrder tie) {
    const int count = min(counters[0] - slot_base, slot_cap);          // flagged queries of this round (uniform)
    if (count <= 0) return;
    __shared__ float xs[FB_TILE][FB_PANEL + 1];
    __shared__ __attribute__((aligned(16))) double qs[FB_PANEL][FB_QG];
    __shared__ uint64_t best[2][FB_QG][FB_KMAX];
    __shared__ uint64_t cand[FB_QG][FB_TILE];
    __shared__ int cand_n[FB_QG];
    __shared__ int cur_s[FB_QG];
    __shared__ int qidx[FB_QG];

    const int tid = threadIdx.x;
    const int splits = gridDim.x;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_split;
    const int64_t r_end = min(n, r_begin + rows_per_split);

    for (int g = blockIdx.y; g * FB_QG < count; g += gridDim.y) {
        if (tid < FB_QG) {
            const int s = g * FB_QG + tid;
            qidx[tid] = s < count ? slots[slot_base + s] : -1;
            cand_n[tid] = 0; cur_s[tid] = 0;
        }
        for (int i = tid; i < FB_QG * FB_KMAX; i += FB_TILE) best[0][i / FB_KMAX][i % FB_KMAX] = ~0ull;
        __syncthreads();
        // panels of FB_PANEL dims, software-pipelined: the next panel's rows and query values are loaded into registers
        // while the current one is multiplied out of LDS (a workgroup has few waves: an exposed load per panel made
        // the scan of one flagged query take ~1 ms)
        const int np = dim / FB_PANEL;
        const int j_q = tid >> 5, c_q = tid & 31;
        float4 nx[FB_TILE * FB_PANEL / 4 / FB_TILE];
        double nqv = 0.0;
        auto load_panel = [&](int64_t r0, int d0) {
#pragma unroll
            for (int u = 0; u < FB_TILE * FB_PANEL / 4 / FB_TILE; ++u) {
                const int idx = tid + FB_TILE * u, rr = idx >> 3, c4 = (idx & 7) * 4;
                const int64_t gr = r0 + rr;
                nx[u] = float4{0.f, 0.f, 0.f, 0.f};
                if (gr < r_end) nx[u] = *(const float4*)(rows + gr * dim + d0 + c4);      // dim % FB_PANEL == 0 (host check)
            }
            const int q = qidx[j_q];
            nqv = q >= 0 ? (double)queries[(int64_t)q * dim + d0 + c_q] : 0.0;
        };
        if (r_begin < r_end) load_panel(r_begin, 0);
        for (int64_t r0 = r_begin; r0 < r_end; r0 += FB_TILE) {
            double acc[FB_QG];
#pragma unroll
            for (int j = 0; j < FB_QG; ++j) acc[j] = 0.0;
            for (int pi = 0; pi < np; ++pi) {
#pragma unroll
                for (int u = 0; u < FB_TILE * FB_PANEL / 4 / FB_TILE; ++u) {
                    const int idx = tid + FB_TILE * u, rr = idx >> 3, c4 = (idx & 7) * 4;
                    xs[rr][c4] = nx[u].x; xs[rr][c4 + 1] = nx[u].y; xs[rr][c4 + 2] = nx[u].z; xs[rr][c4 + 3] = nx[u].w;
                }
                qs[c_q][j_q] = nqv;
                __syncthreads();
                if (pi + 1 < np) load_panel(r0, (pi + 1) * FB_PANEL);
                else if (r0 + FB_TILE < r_end) load_panel(r0 + FB_TILE, 0);
#pragma unroll 4
                for (int c = 0; c < FB_PANEL; ++c) {
                    const double xv = (double)xs[tid][c];
#pragma unroll
                    for (int j = 0; j < FB_QG; ++j) acc[j] += xv * qs[c][j];       // product exact in fp64: index-order chain
                }
                __syncthreads();
            }
            const int64_t row = r0 + tid;
            const uint32_t my_tie = row < r_end ? tie_of(tie, row) : 0u;        // (distance, id) order: TieOrder, vq_common.h
            // rows that beat the query's current k-th key join its candidate list: one LDS atomic per wave and query
            // (a wave-wide ballot places the lanes), not one per row — the first tiles of a scan accept every row
#pragma unroll
            for (int j = 0; j < FB_QG; ++j) {
                if (qidx[j] < 0) continue;                                           // block-uniform
                const uint64_t key = dist_key(1.0f - (float)acc[j], my_tie);
                const bool in = row < r_end && key < best[cur_s[j]][j][k - 1];
                const unsigned long long mask = __ballot(in);
                if (mask == 0) continue;                                             // wave-uniform
                const int lane = tid & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(&cand_n[j], __builtin_popcountll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                if (in) cand[j][base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = key;
            }
            __syncthreads();
            // fold the tile's candidates into the k best so far: rank counting over (list + candidates); keys are
            // unique except the ~0 "empty" entries, which the index tie-break orders
            for (int j = 0; j < FB_QG; ++j) {
                const int cn = cand_n[j];                                            // block-uniform
                if (cn == 0) continue;
                const int cur = cur_s[j], m = k + cn;
                for (int e = tid; e < m; e += FB_TILE) {
                    const uint64_t ke = e < k ? best[cur][j][e] : cand[j][e - k];
                    int rank = 0;
                    for (int f = 0; f < m; ++f) {
                        const uint64_t kf = f < k ? best[cur][j][f] : cand[j][f - k];
                        rank += (kf < ke) || (kf == ke && f < e);
                    }
                    if (rank < k) best[cur ^ 1][j][rank] = ke;
                }
                __syncthreads();
                if (tid == 0) { cur_s[j] = cur ^ 1; cand_n[j] = 0; }
                __syncthreads();
            }
        }
        for (int i = tid; i < FB_QG * k; i += FB_TILE) {
            const int j = i / k, e = i - j * k;
            const int s = g * FB_QG + j;
            if (s < count) partial[((int64_t)s * splits + blockIdx.x) * k + e] = best[cur_s[j]][j][e];
        }
        __syncthreads();
    }
}

// k-way merge of the per-split lists (each ascending): the current head of every list sits in LDS; a round takes the
// smallest head (block minimum; keys are unique: the row is in the key) and only the thread that owns that list
// fetches its next key.  (Re-reading all splits*k keys in each of the k rounds made this 2.3 ms for ONE query.)
__global__ __launch_bounds__(256)
void fallback_merge_kernel(const uint64_t* __restrict__ partial, int splits, int k, const int32_t* __restrict__ slots,
                           const int32_t* __restrict__ counters, int slot_base, int slot_cap,
                           int32_t* __restrict__ ids, float* __restrict__ out_dist, const TieOrder tie) {
    __shared__ uint64_t red[4];
    __shared__ uint64_t head[FB_MAX_SPLITS];
    __shared__ uint8_t pos[FB_MAX_SPLITS];
    const int count = min(counters[0] - slot_base, slot_cap);
    const int tid = threadIdx.x;
    for (int s = blockIdx.x; s < count; s += gridDim.x) {
        const int q = slots[slot_base + s];
        const uint64_t* p = partial + (int64_t)s * splits * k;
        __syncthreads();                                   // the previous query's rounds are over
        for (int l = tid; l < splits; l += 256) { head[l] = p[(int64_t)l * k]; pos[l] = 0; }
        __syncthreads();
        for (int j = 0; j < k; ++j) {
            uint64_t mine = ~0ull; int ml = -1;
            for (int l = tid; l < splits; l += 256) { const uint64_t h = head[l]; if (h < mine) { mine = h; ml = l; } }
            const uint64_t b = block_min_u64(mine, red, tid);
            if (b != ~0ull && mine == b) {                 // exactly one thread: advance the winning list
                const int np = pos[ml] + 1;
                pos[ml] = (uint8_t)np;
                head[ml] = np < k ? p[(int64_t)ml * k + np] : ~0ull;
            }
            if (tid == 0) {
                const int64_t o = (int64_t)q * k + j;
                if (b == ~0ull) { ids[o] = -1; out_dist[o] = __builtin_inff(); }
                else { ids[o] = tie_row(tie, (uint32_t)b); out_dist[o] = key_dist(b); }
            }
            if (b == ~0ull) {                              // every list exhausted (block-uniform)
                for (int jj = j + 1 + tid; jj < k; jj += 256) { ids[(int64_t)q * k + jj] = -1; out_dist[(int64_t)q * k + jj] = __builtin_inff(); }
                break;
            }
        }
    }
}

}  // namespace vq

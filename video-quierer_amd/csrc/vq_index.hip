// vq_index: exact cosine k-NN over a device-resident embedding matrix.
// Replaces HNSWIndex.add / add_batch / search / search_batch / size and backs
// save / load (reference src/indexes/hnsw.py:150-380, 488-528) — SURVEY.md §8a K-rows.
#include "../../include/vq_amd.h"
#include "vq_common.h"
#include "knn_kernels.h"
#include "knn_scan_f16.h"
#include "knn_fallback.h"
#include "knn_scan_deep.h"
#include "knn_scan_fold.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace vq {
int require_init();
enum IdxClass { I_NORMALIZE = 0, I_TO_F16, I_EXACT_DIST, I_SELECT, I_MFMA_SCAN, I_RESCORE };
static const char* kIdxClassNames[VQ_IDX_NCLASS] = {
    "normalize_rows", "rows_to_f16", "exact_dist_f64chain", "select_topk", "scan_f16_mfma_top2", "rescore_verify"};
}  // namespace vq

using namespace vq;

struct vq_index {
    int dim = 0;
    int64_t size = 0, cap = 0;
    float* rows = nullptr;        // fp32 master [cap][dim] (normalised rows, what the reference keeps in .data)
    uint16_t* rows16 = nullptr;   // fp16 scan copy [cap][dim]
    hipStream_t stream = nullptr, own_stream = nullptr;
    std::mutex mu;
    // scratch
    float* d_q = nullptr; int64_t q_cap = 0;          // queries [q_cap][dim]
    float* d_dist = nullptr; int64_t dist_cap = 0;    // exact distances (elements)
    uint64_t* d_partial = nullptr; int64_t partial_cap = 0;   // per-chunk top-k keys
    int32_t* d_ids = nullptr; float* d_out = nullptr; int64_t out_cap = 0;
    float* d_upd = nullptr; int64_t upd_cap = 0;      // vq_index_update_rows: staged rows [n][dim] + their row numbers behind them
    // (distance, id) tie order (vq_index_set_id_ranks): rank of each row's id in the caller's id order + the inverse; rank_n = the
    // number of rows they cover (0 = none set: ties come back in row order).  A search with rank_n != size is refused.
    int32_t* d_rank = nullptr; int32_t* d_rank_inv = nullptr; int64_t rank_cap = 0, rank_n = 0;
    TieOrder tie() const { return rank_n ? TieOrder{d_rank, d_rank_inv} : TieOrder{nullptr, nullptr}; }
    // fp16 scan scratch
    uint16_t* d_q16 = nullptr; int64_t q16_cap = 0;
    uint32_t* d_keys = nullptr; int64_t keys_cap = 0;
    int32_t* d_flags = nullptr; int64_t flags_cap = 0;
    // device-side fallback (knn_fallback.h): flagged query numbers, counters {flagged, proven, rescanned, fallback},
    // per-split top-k lists; the counters travel to pinned memory behind the search, read by last_search_stats
    int32_t* d_slots = nullptr; int64_t slots_cap = 0;
    int32_t* d_counters = nullptr; int32_t* h_counters = nullptr;
    uint64_t* d_fb_partial = nullptr; int64_t fbp_cap = 0;
    int32_t* d_counters_host = nullptr;   // the device's address of h_counters (pinned + mapped): a host-synchronous search lets its kernels write the counters there
    // vq_index_search (host arrays in and out, the reference caller's call): pinned staging for the queries, and a pinned + MAPPED
    // result buffer the kernels write straight into — no device-to-host copy command on the one-query path
    float* h_q = nullptr; int64_t hq_cap = 0;
    char* h_res = nullptr; char* d_res = nullptr; int64_t hres_cap = 0;
    bool fb_deferred = false;      // a host-synchronous fp16 search left its fallback launches to the host (after it has read the flagged count)
    bool stats_pending = false;    // the last search's counters are still on their way to h_counters
    int64_t stats[3] = {0, 0, 0};
    // |row|^2 range of rows added without normalisation (device: min/max fp32 bits); read back lazily by the first
    // search after such an add.  near_unit = the fp16 scan's error bound applies (knn_scan_f16.h scan_eps_unit).
    uint32_t* d_norm_range = nullptr;
    bool norm_dirty = false, near_unit = true;
    float row_norm_max = 1.0f;
    int scan_version = 5;          // $VQ_AMD_SCAN: 5 = deep-prefetch mainloop with the fold spread behind the MFMA clusters (needs dim % 128 == 0), 4 = the same with
                                   // the fold at the row-tile boundary, 2 = 256x256 four-phase, 1 = 128x128; diagnostic builds: 51 / 52 = scan5 without fold / fold not interleaved
    bool no_small_scan = false;    // $VQ_AMD_SCAN_SMALL=0: batches of <= SCAN3_MAX_Q queries also take the MFMA-tile scan (A/B switch)
    bool profiling = false;
    struct Ev { int cls; hipEvent_t a, b; };
    std::vector<Ev> events;
    std::vector<hipEvent_t> pool;
};

namespace {

struct Prof {
    vq_index* x; int cls; hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(vq_index* x) {
        if (!x->pool.empty()) { hipEvent_t ev = x->pool.back(); x->pool.pop_back(); return ev; }
        hipEvent_t ev; (void)hipEventCreate(&ev); return ev;
    }
    Prof(vq_index* i, int c) : x(i), cls(c) {
        if (x->profiling) { a = get(x); b = get(x); (void)hipEventRecord(a, x->stream); }
    }
    ~Prof() { if (x->profiling) { (void)hipEventRecord(b, x->stream); x->events.push_back({cls, a, b}); } }
};

int reserve_rows(vq_index* x, int64_t need) {
    if (need <= x->cap) return 0;
    int64_t ncap = std::max<int64_t>(need, std::max<int64_t>(1024, x->cap * 2));
    ncap = round_up(ncap, SCAN2_RANGE);       // the fp16 scans walk whole 1024/2048-row ranges
    float* nr = nullptr; uint16_t* nh = nullptr;
    hipError_t e = hipMalloc((void**)&nr, (size_t)ncap * x->dim * 4);
    if (e != hipSuccess) return fail(VQ_ERR_OOM, "index: hipMalloc of %lld rows failed: %s", (long long)ncap, hipGetErrorString(e));
    e = hipMalloc((void**)&nh, (size_t)ncap * x->dim * 2);
    if (e != hipSuccess) { (void)hipFree(nr); return fail(VQ_ERR_OOM, "index: hipMalloc (fp16 copy) failed: %s", hipGetErrorString(e)); }
    VQ_HIP(hipMemsetAsync(nh, 0, (size_t)ncap * x->dim * 2, x->stream));   // pad rows of the scan copy stay finite
    if (x->size > 0) {
        VQ_HIP(hipMemcpyAsync(nr, x->rows, (size_t)x->size * x->dim * 4, hipMemcpyDeviceToDevice, x->stream));
        VQ_HIP(hipMemcpyAsync(nh, x->rows16, (size_t)x->size * x->dim * 2, hipMemcpyDeviceToDevice, x->stream));
    }
    VQ_HIP(hipStreamSynchronize(x->stream));
    (void)hipFree(x->rows); (void)hipFree(x->rows16);
    x->rows = nr; x->rows16 = nh; x->cap = ncap;
    return 0;
}

template <class T> int reserve_buf(T*& p, int64_t& cap, int64_t need) {
    if (need <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc((void**)&p, (size_t)need * sizeof(T));
    if (e != hipSuccess) return fail(VQ_ERR_OOM, "index: scratch hipMalloc(%lld) failed: %s", (long long)(need * sizeof(T)), hipGetErrorString(e));
    cap = need;
    return 0;
}

// rows already on the device at x->rows + size*dim
int finish_add(vq_index* x, int64_t n, int normalize) {
    float* dst = x->rows + x->size * x->dim;
    {
        Prof p(x, I_NORMALIZE);
        if (normalize) {
            hipLaunchKernelGGL(normalize_rows_kernel, dim3(cdiv(n, NORM_ROWS)), dim3(NORM_ROWS), 0, x->stream, dst, n, x->dim);
        } else {          // the caller says the rows are unit (HNSWIndex.load, SimpleVideoIndex): measure instead of trusting
            if (!x->d_norm_range) {
                VQ_HIP(hipMalloc((void**)&x->d_norm_range, 8));
                const uint32_t init[2] = {0x3f800000u, 0x3f800000u};            // 1.0f, 1.0f
                VQ_HIP(hipMemcpyAsync(x->d_norm_range, init, 8, hipMemcpyHostToDevice, x->stream));
                VQ_HIP(hipStreamSynchronize(x->stream));                         // `init` is on this stack frame
            }
            hipLaunchKernelGGL(row_norm_range_kernel, dim3(cdiv(n, 4)), dim3(256), 0, x->stream, dst, n, x->dim, x->d_norm_range);
            x->norm_dirty = true;
        }
    }
    {
        Prof p(x, I_TO_F16);
        const int64_t count4 = n * x->dim / 4;
        const int blocks = (int)std::min<int64_t>((count4 + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(rows_to_f16_kernel, dim3(blocks), dim3(256), 0, x->stream, dst,
                           x->rows16 + x->size * x->dim, count4);
    }
    VQ_HIP(hipGetLastError());
    x->size += n;
    return 0;
}

// |row|^2 range -> near_unit / row_norm_max.  Blocks on the stream: called where the entry point blocks anyway (vq_index_add,
// vq_index_update_rows) so that searches after a host add stay asynchronous; the device-side add leaves it to the first search.
int refresh_norm_range(vq_index* x) {
    if (!x->norm_dirty) return 0;
    uint32_t range[2];
    VQ_HIP(hipMemcpyAsync(range, x->d_norm_range, 8, hipMemcpyDeviceToHost, x->stream));
    VQ_HIP(hipStreamSynchronize(x->stream));
    const float lo = __builtin_bit_cast(float, range[0]), hi = __builtin_bit_cast(float, range[1]);
    x->near_unit = lo >= 0.5f && hi <= 2.0f;
    x->row_norm_max = hi > 1.0f ? sqrtf(hi) * 1.0001f : 1.0001f;
    x->norm_dirty = false;
    return 0;
}

// Exact scan: fp64-chain distances for a slice of queries into d_dist, then selection.
int search_exact(vq_index* x, const float* d_queries, int nq, int k, int32_t* d_ids, float* d_dist_out) {
    const int64_t n = x->size;
    const int64_t ld = round_up(n, 64);
    const int64_t budget = (int64_t)128 << 20;                 // 512 MiB of fp32 distances per slice
    int qslice = (int)std::max<int64_t>(32, std::min<int64_t>(nq, budget / ld) / 32 * 32);
    VQ_TRY(reserve_buf(x->d_dist, x->dist_cap, (int64_t)std::min(qslice, (int)round_up(nq, 32)) * ld));
    if (nq <= EDS_MAX_Q && n <= SEL_SMALL_MAX_N && x->dim % 4 == 0 && (size_t)EDS_MAX_Q * x->dim * 8 <= (size_t)96 << 10) {
        // a handful of queries over a small index — the reference caller's one search at a time over a few thousand frames:
        // two short launches (knn_kernels.h)
        const size_t qbytes = (size_t)EDS_MAX_Q * x->dim * 8;
        static std::atomic<size_t> attr_bytes{0};
        if (qbytes > attr_bytes) {
            VQ_HIP(hipFuncSetAttribute((const void*)exact_dist_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qbytes));
            attr_bytes = qbytes;
        }
        {
            Prof p(x, I_EXACT_DIST);
            hipLaunchKernelGGL(exact_dist_small_kernel, dim3(cdiv(n, 64)), dim3(256), qbytes, x->stream, x->rows, n, x->dim, d_queries, nq,
                               x->d_dist, ld);
        }
        {
            Prof p(x, I_SELECT);
            hipLaunchKernelGGL(select_small_kernel, dim3(nq), dim3(256), 0, x->stream, x->d_dist, ld, n, k, d_ids, d_dist_out, x->tie());
        }
        VQ_HIP(hipGetLastError());
        x->stats[0] = 0; x->stats[1] = 0; x->stats[2] = nq;
        x->stats_pending = false;
        return 0;
    }
    const int nchunks = cdiv(n, SEL_CHUNK);
    VQ_TRY(reserve_buf(x->d_partial, x->partial_cap, (int64_t)std::min(qslice, nq) * nchunks * k));
    for (int q0 = 0; q0 < nq; q0 += qslice) {
        const int cur = std::min(qslice, nq - q0);
        {
            Prof p(x, I_EXACT_DIST);
            hipLaunchKernelGGL(exact_dist_kernel, dim3(cdiv(n, 64), cdiv(cur, 32)), dim3(256), 0, x->stream, x->rows, n,
                               x->dim, d_queries + (int64_t)q0 * x->dim, cur, x->d_dist, ld);
        }
        {
            Prof p(x, I_SELECT);
            hipLaunchKernelGGL(select_chunk_kernel, dim3(cur, nchunks), dim3(256), 0, x->stream, x->d_dist, ld, n, k, nchunks,
                               x->d_partial, x->tie());
            hipLaunchKernelGGL(merge_topk_kernel, dim3(cur), dim3(256), 0, x->stream, x->d_partial, nchunks, k,
                               d_ids + (int64_t)q0 * k, d_dist_out + (int64_t)q0 * k, x->tie());
        }
    }
    VQ_HIP(hipGetLastError());
    x->stats[0] = 0; x->stats[1] = 0; x->stats[2] = nq;
    x->stats_pending = false;
    return 0;
}

// fp16 MFMA scan + exact re-score with proof; unproven queries go through search_exact.
static bool large_qpw4() {          // $VQ_AMD_RESCORE_QPW4=1: the four-queries-per-workgroup kernel for small batches with k > 20 too (A/B switch)
    static const bool v = getenv("VQ_AMD_RESCORE_QPW4") && atoi(getenv("VQ_AMD_RESCORE_QPW4")) == 1;
    return v;
}

// The exact redo of the queries whose proof did not close (knn_fallback.h), sized from the device-side flagged count.
void launch_fallback(vq_index* x, const float* d_queries, int nq, int k, int32_t* d_ids, float* d_dist_out, const int32_t* counters) {
    const int64_t n = x->size;
    const int fast_splits = (int)std::max<int64_t>(1, std::min<int64_t>(FB_MAX_SPLITS, cdiv(n, FB_FAST_ROWS)));
    const int64_t fast_rows = round_up(cdiv(n, fast_splits), FB_TILE);
    const int fb_splits = (int)std::max<int64_t>(1, std::min<int64_t>(FB_MAX_SPLITS, cdiv(n, FB_SPLIT_ROWS)));
    const int64_t fb_rows = round_up(cdiv(n, fb_splits), FB_TILE);
    const int64_t fb_cap = std::max<int64_t>(FB_QG, std::min<int64_t>(round_up(nq, FB_QG), ((int64_t)64 << 20) / ((int64_t)fb_splits * k * 8) / FB_QG * FB_QG));
    Prof p(x, I_EXACT_DIST);
    hipLaunchKernelGGL(exact_fallback_kernel, dim3(fast_splits, 1), dim3(FB_TILE), 0, x->stream, x->rows, n, x->dim,
                       d_queries, x->d_slots, counters, 0, FB_FAST_SLOTS, k, fast_rows, x->d_fb_partial, x->tie());
    hipLaunchKernelGGL(fallback_merge_kernel, dim3(FB_FAST_SLOTS), dim3(256), 0, x->stream, x->d_fb_partial, fast_splits, k, x->d_slots,
                       counters, 0, FB_FAST_SLOTS, d_ids, d_dist_out, x->tie());
    for (int64_t base = FB_FAST_SLOTS; base < nq; base += fb_cap) {
        hipLaunchKernelGGL(exact_fallback_kernel, dim3(fb_splits, FB_SLOT_LANES), dim3(FB_TILE), 0, x->stream, x->rows, n, x->dim,
                           d_queries, x->d_slots, counters, (int)base, (int)fb_cap, k, fb_rows, x->d_fb_partial, x->tie());
        hipLaunchKernelGGL(fallback_merge_kernel, dim3(64), dim3(256), 0, x->stream, x->d_fb_partial, fb_splits, k, x->d_slots,
                           counters, (int)base, (int)fb_cap, d_ids, d_dist_out, x->tie());
    }
}

// host_sync: the caller (vq_index_search) waits for the stream anyway, so the outcome counters are written by the kernels into
// host-visible memory and the fallback launches are left to it — it reads the flagged count after its one wait and launches
// them only when there is something to redo (normally nothing: two launches and a copy command fewer per search).
int search_fp16(vq_index* x, const float* d_queries, int nq, int k, int32_t* d_ids, float* d_dist_out, bool host_sync = false) {
    const int64_t n = x->size;
    // small batches (the reference's one-query-at-a-time search, video_search_system.py:297) take the HBM-bound
    // streaming scan; the 256-query MFMA tile is for batches
    const bool small = nq <= SCAN3_MAX_Q && (x->dim == 512 || x->dim == 256 || x->dim == 768) && !x->no_small_scan;
    const int ver = small ? 3 : x->scan_version;                       // 3: streaming, 2: 256x256 phased mainloop, 1: 128x128
    const int nqg3 = nq > SCAN3_QB && x->dim <= 512 ? 2 : 1;           // query groups the streaming scan holds per pass (768-d: 96 + 96 VGPRs for one)
    const int QT = ver == 3 ? SCAN3_QB * nqg3 : ver >= 2 ? SCAN2_QT : SCAN_QT;
    const bool deep = ver == 4 || ver >= 5;                            // the 2048-row-range kernels that share scan2's key layout
    const int RANGE = ver == 3 ? SCAN_STREAM_ROWS : ver >= 2 ? SCAN2_RANGE : SCAN_RANGE;
    const int64_t n_pad = round_up(n, RANGE);
    const int64_t streams = n_pad / SCAN_STREAM_ROWS;
    const int64_t key_budget = (int64_t)1 << 27;                       // 128 Mi (stream,query) pairs = 1 GiB of keys
    int64_t q_chunk = std::max<int64_t>(QT, key_budget / streams / QT * QT);
    q_chunk = std::min<int64_t>(q_chunk, round_up(nq, QT));
    VQ_TRY(reserve_buf(x->d_q16, x->q16_cap, q_chunk * x->dim));
    VQ_TRY(reserve_buf(x->d_keys, x->keys_cap, streams * q_chunk * 2));
    VQ_TRY(reserve_buf(x->d_flags, x->flags_cap, round_up(nq, QT)));
    VQ_TRY(reserve_buf(x->d_slots, x->slots_cap, round_up(nq, 1024)));
    if (!x->d_counters) {
        VQ_HIP(hipMalloc((void**)&x->d_counters, FB_NCOUNTERS * 4));
        VQ_HIP(hipHostMalloc((void**)&x->h_counters, FB_NCOUNTERS * 4, hipHostMallocMapped));
        VQ_HIP(hipHostGetDevicePointer((void**)&x->d_counters_host, x->h_counters, 0));
    }
    int32_t* const counters = host_sync ? x->d_counters_host : x->d_counters;
    // Device-side fallback geometry.  First round: the first FB_FAST_SLOTS flagged queries over fine row splits (many
    // short workgroups: the usual handful of unproven queries is back in ~0.1 ms); bulk rounds: the rest over coarse
    // splits, as many flagged queries per round as 64 MiB of per-split lists hold.  One scratch buffer serves both.
    const int fast_splits = (int)std::max<int64_t>(1, std::min<int64_t>(FB_MAX_SPLITS, cdiv(n, FB_FAST_ROWS)));
    const int64_t fast_rows = round_up(cdiv(n, fast_splits), FB_TILE);
    const int fb_splits = (int)std::max<int64_t>(1, std::min<int64_t>(FB_MAX_SPLITS, cdiv(n, FB_SPLIT_ROWS)));
    const int64_t fb_rows = round_up(cdiv(n, fb_splits), FB_TILE);
    const int64_t fb_cap = std::max<int64_t>(FB_QG, std::min<int64_t>(round_up(nq, FB_QG), ((int64_t)64 << 20) / ((int64_t)fb_splits * k * 8) / FB_QG * FB_QG));
    (void)fast_rows; (void)fb_rows;                     // (launch_fallback derives the same geometry)
    VQ_TRY(reserve_buf(x->d_fb_partial, x->fbp_cap, std::max<int64_t>(fb_cap * fb_splits, (int64_t)FB_FAST_SLOTS * fast_splits) * k));
    const int ranges = (int)(n_pad / RANGE);
    // k in (20, 40] from the streaming scan (the caller's k * 2 for a user k of 11 .. 20): the 64-candidate form of the single-query kernel
    const bool small64 = ver == 3 && k > RV_K_SMALL && k <= RV_K_SMALL64 && x->dim <= 512 && !large_qpw4() && !(getenv("VQ_AMD_RESCORE_SMALL64") && atoi(getenv("VQ_AMD_RESCORE_SMALL64")) == 0);
    if (ver == 3) {
        static std::atomic<bool> attr3_set{false};
        if (!attr3_set) {
            VQ_HIP(hipFuncSetAttribute((const void*)rescore_verify_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (RV_C * (768 + 4) + 768) * 4));
            VQ_HIP(hipFuncSetAttribute((const void*)rescore_verify_small64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (64 * (512 + 4) + 512) * 4));
            attr3_set = true;
        }
    }
    if (ver == 2 || deep) {
        static std::atomic<bool> attr_set{false};
        if (!attr_set) {
            VQ_HIP(hipFuncSetAttribute((const void*)scan2_f16_top2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G2_LDS_BYTES));
            VQ_HIP(hipFuncSetAttribute((const void*)scan4_f16_top2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SCAN4_LDS_BYTES));
            VQ_HIP(hipFuncSetAttribute((const void*)scan5_f16_top2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SCAN4_LDS_BYTES));
#ifdef VQ_DIAG
            VQ_HIP(hipFuncSetAttribute((const void*)scan5_f16_top2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SCAN4_LDS_BYTES));
            VQ_HIP(hipFuncSetAttribute((const void*)scan5_f16_top2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SCAN4_LDS_BYTES));
            VQ_HIP(hipFuncSetAttribute((const void*)scan5_f16_top2_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SCAN4_LDS_BYTES));
#endif
            attr_set = true;
        }
    }
    for (int64_t q0 = 0; q0 < nq; q0 += q_chunk) {
        const int cur = (int)std::min<int64_t>(q_chunk, nq - q0);
        const int64_t q_pad = round_up(cur, QT);
        const int q_tiles = (int)(q_pad / QT);
        const bool fused_q = ver == 3 && nq <= SCAN3_FUSED_MAX_Q;      // the streaming scan rounds the (one to four) queries itself
        if (!fused_q) {
            Prof p(x, I_TO_F16);
            const int64_t total4 = q_pad * x->dim / 4;
            hipLaunchKernelGGL(queries_to_f16_kernel, dim3((int)std::min<int64_t>((total4 + 255) / 256, 2048)), dim3(256), 0,
                               x->stream, d_queries + q0 * x->dim, x->d_q16, cur, q_pad, x->dim);
        }
        {
            Prof p(x, I_MFMA_SCAN);
            if (ver == 3) {
                const dim3 grid(cdiv(streams, 4), q_tiles);
                auto scan3 = x->dim == 768 ? scan3_f16_top2_kernel<24, 1>
                           : x->dim == 512 ? (nqg3 == 2 ? scan3_f16_top2_kernel<16, 2> : scan3_f16_top2_kernel<16, 1>)
                                           : (nqg3 == 2 ? scan3_f16_top2_kernel<8, 2> : scan3_f16_top2_kernel<8, 1>);
                if (fused_q) {
                    auto scan3f = x->dim == 768 ? scan3_f16_top2_kernel<24, 1, true> : x->dim == 512 ? scan3_f16_top2_kernel<16, 1, true>
                                                                                                      : scan3_f16_top2_kernel<8, 1, true>;
                    hipLaunchKernelGGL(scan3f, grid, dim3(256), 0, x->stream, (const uint16_t*)(d_queries + q0 * x->dim), x->rows16, n, streams,
                                       q_pad, x->d_keys, cur);
                } else
                hipLaunchKernelGGL(scan3, grid, dim3(256), 0, x->stream, x->d_q16, x->rows16, n, streams, q_pad, x->d_keys, 0);
            } else if (ver == 2 || deep) {
                const int range_groups = cdiv(ranges, 4), q_groups = cdiv(q_tiles, 8);
                if (ver >= 5) {
                    auto k5 = scan5_f16_top2_kernel<0>;
#ifdef VQ_DIAG
                    if (ver == 51) k5 = scan5_f16_top2_kernel<1>;
                    if (ver == 52) k5 = scan5_f16_top2_kernel<2>;
                    if (ver == 53) k5 = scan5_f16_top2_kernel<0, false>;      // the fold clears its accumulators (no C = 0 MFMAs)
#endif
                    // workgroup -> (row range, query tile) blocking inside an XCD's 32 concurrent workgroups: $VQ_AMD_SCAN_RB = log2 of the
                    // ranges per block (default 2: 4 ranges x 8 query tiles)
                    static const int rb = [] { const char* e = getenv("VQ_AMD_SCAN_RB"); return e ? std::min(5, std::max(0, atoi(e))) : 2; }();   // (thread-safe: searches of different handles run concurrently)
                    const int rg5 = cdiv(ranges, 1 << rb), qg5 = cdiv(q_tiles, 32 >> rb);
                    hipLaunchKernelGGL(k5, dim3(rg5 * qg5 * 32), dim3(G2_THREADS), SCAN4_LDS_BYTES,
                                       x->stream, x->d_q16, x->rows16, x->dim, n, q_tiles, ranges, rg5, q_pad, x->d_keys, x->dim, rb);
                } else if (ver == 4)
                    hipLaunchKernelGGL(scan4_f16_top2_kernel, dim3(range_groups * q_groups * 32), dim3(G2_THREADS), SCAN4_LDS_BYTES,
                                       x->stream, x->d_q16, x->rows16, x->dim, n, q_tiles, ranges, range_groups, q_pad, x->d_keys);
                else
                    hipLaunchKernelGGL(scan2_f16_top2_kernel, dim3(range_groups * q_groups * 32), dim3(G2_THREADS), G2_LDS_BYTES,
                                       x->stream, x->d_q16, x->rows16, x->dim, n, q_tiles, ranges, range_groups, q_pad, x->d_keys);
            }
            else
                hipLaunchKernelGGL(scan_f16_top2_kernel, dim3(q_tiles * ranges), dim3(GEMM_THREADS), 0, x->stream, x->d_q16,
                                   x->rows16, x->dim, n, q_tiles, q_pad, x->d_keys);
        }
        {
            Prof p(x, I_RESCORE);
            const bool large = k > RV_K_SMALL;                  // k in (20, 64]: the wide candidate pool, whatever scan produced the keys
            if (small64)             // k in (20, 40] on the streaming scan's keys: the one-workgroup-per-query kernel with 64 candidates
                hipLaunchKernelGGL(rescore_verify_small64_kernel, dim3(cur), dim3(256), (size_t)(64 * (x->dim + 4) + x->dim) * 4, x->stream, x->d_keys, streams, q_pad, x->rows, n,
                                   x->dim, d_queries + q0 * x->dim, cur, k, d_ids + q0 * k, d_dist_out + q0 * k, x->d_flags + q0,
                                   scan_eps_unit(x->dim) * x->row_norm_max, nq == 1 ? x->d_slots : nullptr, nq == 1 ? counters : nullptr, x->tie());
            else if (large && ver == 3 && !large_qpw4())
                hipLaunchKernelGGL(k > RV_K_MID ? rescore_verify_xlarge1_kernel : rescore_verify_large1_kernel, dim3(cur), dim3(256), 0, x->stream, x->d_keys, streams,
                                   q_pad, x->rows, n, x->dim, d_queries + q0 * x->dim, cur, k, d_ids + q0 * k,
                                   d_dist_out + q0 * k, x->d_flags + q0, 3, scan_eps_unit(x->dim) * x->row_norm_max, x->tie());
            else if (large)
                hipLaunchKernelGGL(k > RV_K_MID ? rescore_verify_xlarge_kernel : rescore_verify_large_kernel, dim3(cdiv(cur, RVL_QPW)), dim3(256), 0, x->stream, x->d_keys, streams,
                                   q_pad, x->rows, n, x->dim, d_queries + q0 * x->dim, cur, k, d_ids + q0 * k,
                                   d_dist_out + q0 * k, x->d_flags + q0, ver == 3 ? 3 : deep ? 2 : ver, scan_eps_unit(x->dim) * x->row_norm_max, x->tie());
            else if (ver == 3)
                hipLaunchKernelGGL(rescore_verify_small_kernel, dim3(cur), dim3(256), (size_t)(RV_C * (x->dim + 4) + x->dim) * 4, x->stream, x->d_keys, streams, q_pad, x->rows, n,
                                   x->dim, d_queries + q0 * x->dim, cur, k, d_ids + q0 * k, d_dist_out + q0 * k, x->d_flags + q0,
                                   scan_eps_unit(x->dim) * x->row_norm_max, nq == 1 ? x->d_slots : nullptr, nq == 1 ? counters : nullptr, x->tie());
            else
            hipLaunchKernelGGL(rescore_verify_kernel, dim3(cdiv(cur, RV_QPW)), dim3(256), 0, x->stream, x->d_keys, streams,
                               q_pad, x->rows, n, x->dim, d_queries + q0 * x->dim, cur, k, d_ids + q0 * k,
                               d_dist_out + q0 * k, x->d_flags + q0, deep ? 2 : ver, scan_eps_unit(x->dim) * x->row_norm_max, x->tie());
        }
    }
    VQ_HIP(hipGetLastError());
    // Queries the proof could not close are redone by the exact scan, on the device: the flags are compacted into a
    // list and the fallback kernels size themselves from its length (all of them leave at once when it is empty), so
    // nothing here waits for the stream.  Rounds beyond the first exist only when more queries could be flagged than
    // one round's scratch holds.
    if (!(ver == 3 && nq == 1 && (k <= RV_K_SMALL || small64)))   // a single query's (small-k) re-score workgroup has written the list and the counters itself
        hipLaunchKernelGGL(collect_flags_kernel, dim3(1), dim3(1024), 0, x->stream, x->d_flags, nq, x->d_slots, counters);
    VQ_HIP(hipGetLastError());
    if (host_sync) { x->fb_deferred = true; x->stats_pending = false; return 0; }
    launch_fallback(x, d_queries, nq, k, d_ids, d_dist_out, counters);
    VQ_HIP(hipGetLastError());
    VQ_HIP(hipMemcpyAsync(x->h_counters, x->d_counters, FB_NCOUNTERS * 4, hipMemcpyDeviceToHost, x->stream));
    x->stats_pending = true;
    return 0;
}

int search_dispatch(vq_index* x, const float* d_queries, int nq, int k, int mode, int32_t* d_ids, float* d_dist, bool host_sync = false) {
    VQ_CHECK(mode >= 0 && mode <= 2, "vq_index_search: mode %d unknown", mode);
    VQ_CHECK(x->rank_n == 0 || x->rank_n == x->size, "vq_index_search: the id ranks cover %lld rows, the index holds %lld "
             "(call vq_index_set_id_ranks again after adding rows, or clear them)", (long long)x->rank_n, (long long)x->size);
    // rows were added un-normalised ON THE DEVICE since the last look (vq_index_add_device: the one add that does not block):
    // is the matrix still near-unit?  This is the only place a search waits for its stream.
    if (mode != 1) VQ_TRY(refresh_norm_range(x));
    const bool fp16_ok = x->dim % GEMM_BK == 0 && k <= RV_K_MAX && x->size >= 1 && x->near_unit;
    if (mode == 2) VQ_CHECK(fp16_ok, "vq_index_search: fp16 scan needs dim %% 64 == 0, k <= %d and near-unit rows "
                                     "(0.5 <= |row|^2 <= 2; rows added with normalize=0 are measured)", RV_K_MAX);
    // auto: the MFMA scan pays once the matrix is large enough to amortise its fixed costs
    const bool use_fp16 = mode == 2 || (mode == 0 && fp16_ok && x->size >= 16384);
    return use_fp16 ? search_fp16(x, d_queries, nq, k, d_ids, d_dist, host_sync) : search_exact(x, d_queries, nq, k, d_ids, d_dist);
}

}  // namespace

// vq_comm.hip: this rank's part of a row-sharded search, on the index's stream (returned so that the exchange is
// enqueued behind it).
namespace vq {
hipStream_t index_stream(vq_index* x) {
    std::lock_guard<std::mutex> lk(x->mu);
    return x->stream;
}
int index_search_local(vq_index* x, const float* d_queries, int nq, int k, int mode, int32_t* d_ids, float* d_dist,
                       hipStream_t* stream_out, int64_t* size_out) {
    std::lock_guard<std::mutex> lk(x->mu);
    *stream_out = x->stream;
    *size_out = x->size;
    if (x->size == 0) {
        const int64_t count = (int64_t)nq * k;
        hipLaunchKernelGGL(fill_no_result_kernel, dim3(cdiv(count, 256)), dim3(256), 0, x->stream, d_ids, d_dist, count);
        VQ_HIP(hipGetLastError());
        return 0;
    }
    return search_dispatch(x, d_queries, nq, k, mode, d_ids, d_dist);
}
}  // namespace vq

extern "C" {

int vq_index_create(int dim, vq_index** out) {
    VQ_TRY(require_init());
    VQ_CHECK(out && dim > 0 && dim % 4 == 0 && dim <= 4096, "vq_index_create: dim %d must be a positive multiple of 4", dim);
    vq_index* x = new vq_index();
    x->dim = dim;
    if (const char* sv = getenv("VQ_AMD_SCAN")) { const int v = atoi(sv); x->scan_version = (v == 1 || v == 2 || v == 4 || v == 51 || v == 52 || v == 53) ? v : 5; }
    if (dim % 128 != 0) x->scan_version = 1;
    if (const char* ss = getenv("VQ_AMD_SCAN_SMALL")) x->no_small_scan = atoi(ss) == 0;
    hipError_t e = hipStreamCreateWithFlags(&x->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete x; return fail(VQ_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
    x->stream = x->own_stream;
    *out = x;
    return 0;
}

int vq_index_destroy(vq_index* x) {
    if (!x) return 0;
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->own_stream) (void)hipStreamDestroy(x->own_stream);
    for (auto& ev : x->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto ev : x->pool) (void)hipEventDestroy(ev);
    (void)hipFree(x->rows); (void)hipFree(x->rows16); (void)hipFree(x->d_q); (void)hipFree(x->d_dist);
    (void)hipFree(x->d_ids); (void)hipFree(x->d_out); (void)hipFree(x->d_partial);
    (void)hipFree(x->d_q16); (void)hipFree(x->d_keys); (void)hipFree(x->d_flags); (void)hipFree(x->d_slots);
    (void)hipFree(x->d_counters); (void)hipFree(x->d_fb_partial); (void)hipFree(x->d_upd);
    if (x->h_counters) (void)hipHostFree(x->h_counters);
    if (x->h_q) (void)hipHostFree(x->h_q);
    if (x->h_res) (void)hipHostFree(x->h_res);
    (void)hipFree(x->d_norm_range); (void)hipFree(x->d_rank);
    delete x;
    return 0;
}

int vq_index_size(vq_index* x, int64_t* n) {
    VQ_CHECK(x && n, "vq_index_size: null argument");
    *n = x->size;
    return 0;
}

int vq_index_clear(vq_index* x) {
    VQ_CHECK(x, "vq_index_clear: null handle");
    std::lock_guard<std::mutex> lk(x->mu);
    x->size = 0;
    if (x->d_norm_range) {
        const uint32_t init[2] = {0x3f800000u, 0x3f800000u};
        VQ_HIP(hipMemcpyAsync(x->d_norm_range, init, 8, hipMemcpyHostToDevice, x->stream));
        VQ_HIP(hipStreamSynchronize(x->stream));
    }
    x->norm_dirty = false; x->near_unit = true; x->row_norm_max = 1.0f;
    x->rank_n = 0;
    return 0;
}

int vq_index_set_id_ranks(vq_index* x, const int32_t* rank_of_row, int64_t n) {
    VQ_TRY(require_init());
    VQ_CHECK(x && n >= 0 && (n == 0 || rank_of_row), "vq_index_set_id_ranks: bad argument");
    std::lock_guard<std::mutex> lk(x->mu);
    if (n == 0) { x->rank_n = 0; return 0; }                     // back to row order
    VQ_CHECK(n == x->size, "vq_index_set_id_ranks: %lld ranks for an index of %lld rows", (long long)n, (long long)x->size);
    // a permutation of 0..n-1, checked here: the kernels index two arrays with these values
    std::vector<int32_t> inv((size_t)n, -1);
    for (int64_t r = 0; r < n; ++r) {
        const int32_t t = rank_of_row[r];
        VQ_CHECK(t >= 0 && t < n && inv[(size_t)t] < 0, "vq_index_set_id_ranks: rank_of_row is not a permutation of 0..%lld "
                 "(row %lld holds %d)", (long long)n - 1, (long long)r, (int)t);
        inv[(size_t)t] = (int32_t)r;
    }
    if (n > x->rank_cap) {
        VQ_HIP(hipStreamSynchronize(x->stream));                   // a search in flight may still read the old arrays
        (void)hipFree(x->d_rank); x->d_rank = nullptr; x->d_rank_inv = nullptr; x->rank_cap = 0; x->rank_n = 0;
        const int64_t cap = round_up(std::max<int64_t>(n, x->cap), 1024);
        hipError_t e = hipMalloc((void**)&x->d_rank, (size_t)cap * 8);
        if (e != hipSuccess) return fail(VQ_ERR_OOM, "vq_index_set_id_ranks: hipMalloc failed: %s", hipGetErrorString(e));
        x->d_rank_inv = x->d_rank + cap;
        x->rank_cap = cap;
    }
    VQ_HIP(hipMemcpyAsync(x->d_rank, rank_of_row, (size_t)n * 4, hipMemcpyHostToDevice, x->stream));
    VQ_HIP(hipMemcpyAsync(x->d_rank_inv, inv.data(), (size_t)n * 4, hipMemcpyHostToDevice, x->stream));
    VQ_HIP(hipStreamSynchronize(x->stream));                       // `inv` is this frame's, `rank_of_row` the caller's
    x->rank_n = n;
    return 0;
}

int vq_index_add(vq_index* x, const float* rows, int64_t n, int normalize) {
    VQ_TRY(require_init());
    VQ_CHECK(x && n >= 0 && (n == 0 || rows), "vq_index_add: bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(x->mu);
    VQ_CHECK(x->size + n < ((int64_t)1 << 31), "vq_index_add: more than 2^31 rows");
    VQ_TRY(reserve_rows(x, x->size + n));
    VQ_HIP(hipMemcpyAsync(x->rows + x->size * x->dim, rows, (size_t)n * x->dim * 4, hipMemcpyHostToDevice, x->stream));
    VQ_TRY(finish_add(x, n, normalize));
    VQ_TRY(refresh_norm_range(x));                // this call blocks anyway: searches that follow need not
    VQ_HIP(hipStreamSynchronize(x->stream));
    return 0;
}

int vq_index_add_device(vq_index* x, const void* d_rows, int64_t n, int normalize) {
    VQ_TRY(require_init());
    VQ_CHECK(x && n >= 0 && (n == 0 || d_rows), "vq_index_add_device: bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(x->mu);
    VQ_CHECK(x->size + n < ((int64_t)1 << 31), "vq_index_add_device: more than 2^31 rows");
    VQ_TRY(reserve_rows(x, x->size + n));
    VQ_HIP(hipMemcpyAsync(x->rows + x->size * x->dim, d_rows, (size_t)n * x->dim * 4, hipMemcpyDeviceToDevice, x->stream));
    return finish_add(x, n, normalize);
}

int vq_index_update_rows(vq_index* x, const float* rows, const int64_t* row_numbers, int64_t n, int normalize) {
    VQ_TRY(require_init());
    VQ_CHECK(x && n >= 0 && (n == 0 || (rows && row_numbers)), "vq_index_update_rows: bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(x->mu);
    // the reference assigns in call order (hnsw.py:160), so the LAST update of a row is the one that stays: keep that one
    std::vector<int64_t> keep;            // indices into rows / row_numbers, in order
    {
        std::vector<std::pair<int64_t, int64_t>> last;          // (row number, index of its last update)
        last.reserve((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            VQ_CHECK(row_numbers[i] >= 0 && row_numbers[i] < x->size, "vq_index_update_rows: row %lld outside [0, %lld)",
                     (long long)row_numbers[i], (long long)x->size);
            last.emplace_back(row_numbers[i], i);
        }
        std::stable_sort(last.begin(), last.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (size_t i = 0; i < last.size(); ++i)
            if (i + 1 == last.size() || last[i + 1].first != last[i].first) keep.push_back(last[i].second);
        std::sort(keep.begin(), keep.end());
    }
    const int64_t m = (int64_t)keep.size();
    const int64_t row_bytes = (int64_t)x->dim * 4;
    VQ_TRY(reserve_buf(x->d_upd, x->upd_cap, m * x->dim + m * 2 + 4));
    int64_t* d_rn = (int64_t*)(x->d_upd + round_up(m * x->dim, 2));       // 8-byte aligned behind the rows
    std::vector<int64_t> rn((size_t)m);
    // `rn` and the caller's `rows` feed asynchronous copies: whichever way this function is left (every VQ_HIP / VQ_TRY below
    // returns early on error), the stream is drained before `rn` dies (declared after it: destroyed first)
    struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{x->stream};
    if (m == n) {
        VQ_HIP(hipMemcpyAsync(x->d_upd, rows, (size_t)(n * row_bytes), hipMemcpyHostToDevice, x->stream));
        for (int64_t i = 0; i < m; ++i) rn[(size_t)i] = row_numbers[i];
    } else {
        for (int64_t i = 0; i < m; ++i) {
            VQ_HIP(hipMemcpyAsync(x->d_upd + i * x->dim, rows + keep[(size_t)i] * x->dim, (size_t)row_bytes, hipMemcpyHostToDevice, x->stream));
            rn[(size_t)i] = row_numbers[keep[(size_t)i]];
        }
    }
    VQ_HIP(hipMemcpyAsync(d_rn, rn.data(), (size_t)m * 8, hipMemcpyHostToDevice, x->stream));
    {
        Prof p(x, I_NORMALIZE);
        if (normalize) {
            hipLaunchKernelGGL(normalize_rows_kernel, dim3(cdiv(m, NORM_ROWS)), dim3(NORM_ROWS), 0, x->stream, x->d_upd, m, x->dim);
        } else {          // measured, not trusted (finish_add); the range only widens: a replaced row's old norm stays covered
            if (!x->d_norm_range) {
                VQ_HIP(hipMalloc((void**)&x->d_norm_range, 8));
                const uint32_t init[2] = {0x3f800000u, 0x3f800000u};
                VQ_HIP(hipMemcpyAsync(x->d_norm_range, init, 8, hipMemcpyHostToDevice, x->stream));
                VQ_HIP(hipStreamSynchronize(x->stream));
            }
            hipLaunchKernelGGL(row_norm_range_kernel, dim3(cdiv(m, 4)), dim3(256), 0, x->stream, x->d_upd, m, x->dim, x->d_norm_range);
            x->norm_dirty = true;
        }
    }
    {
        Prof p(x, I_TO_F16);
        const int64_t count4 = m * x->dim / 4;
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((int)std::min<int64_t>((count4 + 255) / 256, 2048)), dim3(256), 0, x->stream,
                           x->d_upd, d_rn, m, x->dim, x->rows, x->rows16);
    }
    VQ_HIP(hipGetLastError());
    VQ_TRY(refresh_norm_range(x));
    VQ_HIP(hipStreamSynchronize(x->stream));      // `rows`, `rn` are the caller's / this frame's
    return 0;
}

int vq_index_search_device(vq_index* x, const void* d_queries, int nq, int k, int mode, void* d_ids, void* d_dist) {
    VQ_TRY(require_init());
    VQ_CHECK(x && nq >= 0 && k > 0 && k <= 1024 && (nq == 0 || (d_queries && d_ids && d_dist)), "vq_index_search_device: bad argument");
    if (nq == 0) return 0;
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->size == 0) {                       // empty index: no candidates, as vq_index_search reports it
        const int64_t count = (int64_t)nq * k;
        hipLaunchKernelGGL(fill_no_result_kernel, dim3(cdiv(count, 256)), dim3(256), 0, x->stream, (int32_t*)d_ids, (float*)d_dist, count);
        VQ_HIP(hipGetLastError());
        return 0;
    }
    return search_dispatch(x, (const float*)d_queries, nq, k, mode, (int32_t*)d_ids, (float*)d_dist);
}

int vq_index_search(vq_index* x, const float* queries, int nq, int k, int mode, int32_t* ids, float* dist) {
    VQ_TRY(require_init());
    VQ_CHECK(x && nq >= 0 && k > 0 && k <= 1024 && (nq == 0 || (queries && ids && dist)), "vq_index_search: bad argument");
    if (nq == 0) return 0;
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->size == 0) {                       // empty index: no candidates (hnsw.py:243-244 returns [])
        for (int64_t i = 0; i < (int64_t)nq * k; ++i) { ids[i] = -1; dist[i] = __builtin_inff(); }
        return 0;
    }
    VQ_TRY(reserve_buf(x->d_q, x->q_cap, (int64_t)nq * x->dim));
    const int64_t q_bytes = (int64_t)nq * x->dim * 4, res_bytes = (int64_t)nq * k * 8;
    static const bool host_fast = !(getenv("VQ_AMD_HOST_FAST") && atoi(getenv("VQ_AMD_HOST_FAST")) == 0);      // A/B switch
    if (host_fast && q_bytes <= ((int64_t)256 << 10) && res_bytes <= ((int64_t)16 << 10)) {      // (larger results: scattered 4-byte stores across PCIe lose to one copy)
        // The reference caller's call (one query, k * 2 results: video_search_system.py:297) and small batches: the query goes up
        // from pinned staging, the kernels write ids | distances straight into pinned, device-mapped host memory, and the ONE
        // wait below is the only host/device round trip — no copy-back commands, no fallback launches unless a query was flagged.
        if (q_bytes > x->hq_cap) {
            if (x->h_q) (void)hipHostFree(x->h_q);
            x->h_q = nullptr; x->hq_cap = 0;
            VQ_HIP(hipHostMalloc((void**)&x->h_q, (size_t)std::max<int64_t>(q_bytes, 64 << 10)));
            x->hq_cap = std::max<int64_t>(q_bytes, 64 << 10);
        }
        if (res_bytes > x->hres_cap) {
            VQ_HIP(hipStreamSynchronize(x->stream));
            if (x->h_res) (void)hipHostFree(x->h_res);
            x->h_res = nullptr; x->hres_cap = 0;
            VQ_HIP(hipHostMalloc((void**)&x->h_res, (size_t)std::max<int64_t>(res_bytes, 64 << 10), hipHostMallocMapped));
            VQ_HIP(hipHostGetDevicePointer((void**)&x->d_res, x->h_res, 0));
            x->hres_cap = std::max<int64_t>(res_bytes, 64 << 10);
        }
        memcpy(x->h_q, queries, (size_t)q_bytes);
        VQ_HIP(hipMemcpyAsync(x->d_q, x->h_q, (size_t)q_bytes, hipMemcpyHostToDevice, x->stream));
        int32_t* r_ids = (int32_t*)x->d_res;
        float* r_dist = (float*)(x->d_res + (size_t)nq * k * 4);
        x->fb_deferred = false;
        VQ_TRY(search_dispatch(x, x->d_q, nq, k, mode, r_ids, r_dist, true));
        VQ_HIP(hipStreamSynchronize(x->stream));
        if (x->fb_deferred) {
            x->fb_deferred = false;
            for (int i = 0; i < 3; ++i) x->stats[i] = x->h_counters[1 + i];
            if (x->h_counters[0] > 0) {                      // some proof did not close: the exact redo, then one more wait
                launch_fallback(x, x->d_q, nq, k, r_ids, r_dist, x->d_counters_host);
                VQ_HIP(hipGetLastError());
                VQ_HIP(hipStreamSynchronize(x->stream));
            }
        }
        memcpy(ids, x->h_res, (size_t)nq * k * 4);
        memcpy(dist, x->h_res + (size_t)nq * k * 4, (size_t)nq * k * 4);
        return 0;
    }
    if ((int64_t)nq * k > x->out_cap) {
        int64_t c1 = x->out_cap, c2 = x->out_cap;
        VQ_TRY(reserve_buf(x->d_ids, c1, (int64_t)nq * k));
        VQ_TRY(reserve_buf(x->d_out, c2, (int64_t)nq * k));
        x->out_cap = (int64_t)nq * k;
    }
    VQ_HIP(hipMemcpyAsync(x->d_q, queries, (size_t)nq * x->dim * 4, hipMemcpyHostToDevice, x->stream));
    VQ_TRY(search_dispatch(x, x->d_q, nq, k, mode, x->d_ids, x->d_out));
    VQ_HIP(hipMemcpyAsync(ids, x->d_ids, (size_t)nq * k * 4, hipMemcpyDeviceToHost, x->stream));
    VQ_HIP(hipMemcpyAsync(dist, x->d_out, (size_t)nq * k * 4, hipMemcpyDeviceToHost, x->stream));
    VQ_HIP(hipStreamSynchronize(x->stream));
    return 0;
}

int vq_index_synchronize(vq_index* x) {
    VQ_CHECK(x, "vq_index_synchronize: null handle");
    VQ_HIP(hipStreamSynchronize(x->stream));
    return 0;
}

int vq_index_set_stream(vq_index* x, void* hip_stream) {
    VQ_CHECK(x, "vq_index_set_stream: null handle");
    std::lock_guard<std::mutex> lk(x->mu);
    VQ_HIP(hipStreamSynchronize(x->stream));
    x->stream = hip_stream ? (hipStream_t)hip_stream : x->own_stream;
    return 0;
}

int vq_index_export(vq_index* x, float* rows) {
    VQ_TRY(require_init());
    VQ_CHECK(x && (x->size == 0 || rows), "vq_index_export: null argument");
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->size == 0) return 0;
    VQ_HIP(hipMemcpyAsync(rows, x->rows, (size_t)x->size * x->dim * 4, hipMemcpyDeviceToHost, x->stream));
    VQ_HIP(hipStreamSynchronize(x->stream));
    return 0;
}

int vq_index_read_rows(vq_index* x, const int64_t* row_numbers, int64_t n, float* out) {
    VQ_TRY(require_init());
    VQ_CHECK(x && n >= 0 && (n == 0 || (row_numbers && out)), "vq_index_read_rows: bad argument");
    std::lock_guard<std::mutex> lk(x->mu);
    for (int64_t i = 0; i < n; ++i) {
        VQ_CHECK(row_numbers[i] >= 0 && row_numbers[i] < x->size, "vq_index_read_rows: row %lld outside [0, %lld)",
                 (long long)row_numbers[i], (long long)x->size);
        VQ_HIP(hipMemcpyAsync(out + i * x->dim, x->rows + row_numbers[i] * x->dim, (size_t)x->dim * 4, hipMemcpyDeviceToHost, x->stream));
    }
    VQ_HIP(hipStreamSynchronize(x->stream));
    return 0;
}

int vq_index_profile_begin(vq_index* x) {
    VQ_CHECK(x, "vq_index_profile_begin: null handle");
    std::lock_guard<std::mutex> lk(x->mu);
    VQ_HIP(hipStreamSynchronize(x->stream));
    for (auto& ev : x->events) { x->pool.push_back(ev.a); x->pool.push_back(ev.b); }
    x->events.clear();
    x->profiling = true;
    return 0;
}

int vq_index_profile_end(vq_index* x, float* ms, int* launches) {
    VQ_CHECK(x && ms && launches, "vq_index_profile_end: null argument");
    std::lock_guard<std::mutex> lk(x->mu);
    x->profiling = false;
    VQ_HIP(hipStreamSynchronize(x->stream));
    for (int i = 0; i < VQ_IDX_NCLASS; ++i) { ms[i] = 0.f; launches[i] = 0; }
    for (auto& ev : x->events) {
        float t = 0.f;
        VQ_HIP(hipEventElapsedTime(&t, ev.a, ev.b));
        ms[ev.cls] += t; launches[ev.cls] += 1;
        x->pool.push_back(ev.a); x->pool.push_back(ev.b);
    }
    x->events.clear();
    return 0;
}

const char* vq_index_profile_class_name(int cls) {
    return (cls >= 0 && cls < VQ_IDX_NCLASS) ? kIdxClassNames[cls] : "";
}

int vq_index_last_search_stats(vq_index* x, int64_t* stats) {
    VQ_CHECK(x && stats, "vq_index_last_search_stats: null argument");
    std::lock_guard<std::mutex> lk(x->mu);
    if (x->stats_pending) {                    // the fp16 path's counters follow the search on its stream
        VQ_HIP(hipStreamSynchronize(x->stream));
        for (int i = 0; i < 3; ++i) x->stats[i] = x->h_counters[1 + i];
        x->stats_pending = false;
    }
    for (int i = 0; i < 3; ++i) stats[i] = x->stats[i];
    return 0;
}

}  // extern "C"

#ifdef VQ_GEMM_TOWER_STAMPS
// `make STAMPS=1` only: the phase boundaries rescore_verify_small_kernel's workgroup 0 stamped in its last launch (scripts/rescore_stamps.py)
extern "C" int vq_debug_dump_rescore_stamps(void) {
    unsigned long long h[16];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(vq::g_rs_stamps), sizeof(h));
    static const char* names[9] = {"keys pass A: thread maxima", "wave 8th largest, |q|^2, barrier", "keys pass B: collect, rank", "first-pass rows -> LDS", "query -> LDS, barrier",
                                   "fp64 chains, k-th distance (+ 2nd pass)", "verdict", "rescans", "top-k out"};
    for (int i = 0; i < 9; ++i) fprintf(stderr, "RS_STAMP %-32s %8llu cycles\n", names[i], h[i + 1] - h[i]);
    fprintf(stderr, "RS_STAMP %-32s %8llu cycles\n", "total", h[9] - h[0]);
    return 0;
}
#endif

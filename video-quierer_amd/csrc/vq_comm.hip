// vq_comm: the path's two exchange steps over RCCL (xGMI), one process per GPU.
//
// The reference has no communication layer (single process, single device); the
// north star adds frame-sharded ingest with an all-gather of the per-shard embeddings before
// indexing (reference call sites it stands in front of: src/video_search_system.py:152-181) and
// a row-sharded search whose only exchange is the per-shard top-k (src/video_search_system.py:297
// on each shard, then a k-way merge in the (distance, id) order of src/indexes/hnsw.py:269).
//
// librccl is resolved at run time (dlopen by soname: in a process that already holds a copy —
// torch ships one — that copy is used), so single-GPU users never load it.
#include "../../include/vq_amd.h"
#include "vq_common.h"

#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>
#include <vector>

namespace vq {
int require_init();
int index_search_local(vq_index* x, const float* d_queries, int nq, int k, int mode, int32_t* d_ids, float* d_dist,
                       hipStream_t* stream_out, int64_t* size_out);

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
        r.GetVersion = (decltype(r.GetVersion))dlsym(r.so, "ncclGetVersion");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
    });
    const bool ok = r.so && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
    return ok ? &r : nullptr;
}

#define VQ_NCCL(expr)                                                                               \
    do {                                                                                            \
        ncclResult_t r__ = (expr);                                                                  \
        if (r__ != ncclSuccess)                                                                     \
            return ::vq::fail(VQ_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(r__));     \
    } while (0)

// local shard result -> exchange keys: key = dist_key(distance, GLOBAL row id); an empty slot (id -1) stays the
// largest possible key, so it loses every comparison in the merge
__global__ void pack_keys_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dist, int64_t count,
                                 int64_t row_offset, uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int32_t id = ids[i];
    keys[i] = id < 0 ? ~0ull : dist_key(dist[i], (uint32_t)(id + row_offset));
}

// [W][Q][k] keys -> the k smallest per query, ascending = (distance asc, global id asc), hnsw.py:269.
// One 64-lane wave per query; W*k <= 1024 candidates are ranked by counting (keys are unique except for
// empty slots, which are ordered by position so that ranks stay a permutation).
constexpr int MERGE_MAX = 1024;
__global__ __launch_bounds__(256)
void merge_keys_kernel(const uint64_t* __restrict__ keys, int world, int nq, int k, int32_t* __restrict__ out_ids,
                       float* __restrict__ out_dist) {
    __shared__ uint64_t cand[4][MERGE_MAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    const int n = world * k;
    if (q < nq)
        for (int i = lane; i < n; i += 64) {
            const int r = i / k, j = i - r * k;
            cand[w][i] = keys[((size_t)r * nq + q) * k + j];
        }
    __syncthreads();
    if (q >= nq) return;
    for (int i = lane; i < n; i += 64) {
        const uint64_t ki = cand[w][i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const uint64_t kj = cand[w][j];
            rank += (kj < ki) || (kj == ki && j < i);
        }
        if (rank < k) {
            const size_t o = (size_t)q * k + rank;
            if (ki == ~0ull) { out_ids[o] = -1; out_dist[o] = __builtin_inff(); }
            else { out_ids[o] = (int32_t)(uint32_t)ki; out_dist[o] = key_dist(ki); }
        }
    }
}

template <class T> int grow(T*& p, int64_t& cap, int64_t need) {
    if (need <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    hipError_t e = hipMalloc((void**)&p, (size_t)need * sizeof(T));
    if (e != hipSuccess) return fail(VQ_ERR_OOM, "vq_comm: scratch hipMalloc(%lld) failed: %s", (long long)(need * sizeof(T)), hipGetErrorString(e));
    cap = need;
    return 0;
}

}  // namespace
}  // namespace vq

using namespace vq;

struct vq_comm {
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    std::mutex mu;
    // scratch of the sharded search: local result, exchange keys (mine, everyone's), padded gather buffer
    int32_t* d_lids = nullptr; int64_t lids_cap = 0;
    float* d_ldist = nullptr; int64_t ldist_cap = 0;
    uint64_t* d_keys = nullptr; int64_t keys_cap = 0;
    uint64_t* d_all = nullptr; int64_t all_cap = 0;
    float* d_pad = nullptr; int64_t pad_cap = 0;
    float* d_gath = nullptr; int64_t gath_cap = 0;
};

extern "C" {

int vq_comm_unique_id(void* out_id, int bytes) {
    VQ_CHECK(out_id && bytes >= (int)sizeof(ncclUniqueId), "vq_comm_unique_id: need a %d-byte buffer", (int)sizeof(ncclUniqueId));
    Rccl* r = rccl();
    if (!r) return fail(VQ_ERR_STATE, "vq_comm: librccl.so.1 could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");
    ncclUniqueId id;
    VQ_NCCL(r->GetUniqueId(&id));
    memcpy(out_id, &id, sizeof(id));
    return 0;
}

int vq_comm_init(int rank, int world, const void* unique_id, vq_comm** out) {
    VQ_TRY(require_init());             // the communicator belongs to the device this process is bound to
    VQ_CHECK(out && unique_id && world >= 1 && rank >= 0 && rank < world, "vq_comm_init: bad rank %d / world %d", rank, world);
    Rccl* r = rccl();
    if (!r) return fail(VQ_ERR_STATE, "vq_comm: librccl.so.1 could not be loaded");
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    vq_comm* c = new vq_comm();
    c->rank = rank; c->world = world;
    ncclResult_t nr = r->CommInitRank(&c->comm, world, id, rank);
    if (nr != ncclSuccess) { delete c; return fail(VQ_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, r->GetErrorString(nr)); }
    *out = c;
    return 0;
}

int vq_comm_destroy(vq_comm* c) {
    if (!c) return 0;
    (void)hipDeviceSynchronize();
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    (void)hipFree(c->d_lids); (void)hipFree(c->d_ldist); (void)hipFree(c->d_keys); (void)hipFree(c->d_all);
    (void)hipFree(c->d_pad); (void)hipFree(c->d_gath);
    delete c;
    return 0;
}

int vq_comm_info(vq_comm* c, int* rank, int* world, int* rccl_version) {
    VQ_CHECK(c, "vq_comm_info: null handle");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (rccl_version) { *rccl_version = 0; if (rccl() && rccl()->GetVersion) (void)rccl()->GetVersion(rccl_version); }
    return 0;
}

// Ingest exchange: every rank contributes counts[rank] rows of `dim` floats; every rank receives all of them in
// rank (= frame) order.  Equal counts: one ncclAllGather straight into d_out.  Ragged: ranks pad to the largest
// count, gather into scratch, and the valid prefixes are compacted into d_out.
int vq_allgather_rows(vq_comm* c, const void* d_local, const int64_t* counts, int dim, void* d_out, void* hip_stream) {
    VQ_TRY(require_init());
    VQ_CHECK(c && counts && dim > 0 && d_out, "vq_allgather_rows: bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    std::lock_guard<std::mutex> lk(c->mu);
    int64_t mx = 0, total = 0; bool ragged = false;
    for (int r = 0; r < c->world; ++r) {
        VQ_CHECK(counts[r] >= 0, "vq_allgather_rows: negative count");
        mx = std::max(mx, counts[r]); total += counts[r]; ragged |= counts[r] != counts[0];
    }
    if (total == 0) return 0;
    VQ_CHECK(counts[c->rank] == 0 || d_local, "vq_allgather_rows: null local rows");
    Rccl* r = rccl();
    if (!ragged) {
        VQ_NCCL(r->AllGather(d_local, d_out, (size_t)mx * dim, ncclFloat, c->comm, st));
        return 0;
    }
    VQ_TRY(grow(c->d_pad, c->pad_cap, mx * dim));
    VQ_TRY(grow(c->d_gath, c->gath_cap, mx * dim * c->world));
    VQ_HIP(hipMemsetAsync(c->d_pad, 0, (size_t)mx * dim * 4, st));
    if (counts[c->rank] > 0)
        VQ_HIP(hipMemcpyAsync(c->d_pad, d_local, (size_t)counts[c->rank] * dim * 4, hipMemcpyDeviceToDevice, st));
    VQ_NCCL(r->AllGather(c->d_pad, c->d_gath, (size_t)mx * dim, ncclFloat, c->comm, st));
    int64_t off = 0;
    for (int rr = 0; rr < c->world; ++rr) {
        if (counts[rr] > 0)
            VQ_HIP(hipMemcpyAsync((float*)d_out + off * dim, c->d_gath + (int64_t)rr * mx * dim, (size_t)counts[rr] * dim * 4,
                                  hipMemcpyDeviceToDevice, st));
        off += counts[rr];
    }
    return 0;
}

// The merge step on its own: [world][nq][k] shard results (GLOBAL ids, -1 = empty) -> exact [nq][k].
int vq_merge_topk_device(const void* d_all_ids, const void* d_all_dist, int world, int nq, int k, void* d_ids, void* d_dist,
                         void* hip_stream) {
    VQ_TRY(require_init());
    VQ_CHECK(d_all_ids && d_all_dist && d_ids && d_dist && world >= 1 && nq >= 0 && k > 0 && (int64_t)world * k <= MERGE_MAX,
             "vq_merge_topk_device: bad argument (world*k must be <= %d)", MERGE_MAX);
    if (nq == 0) return 0;
    hipStream_t st = (hipStream_t)hip_stream;
    const int64_t count = (int64_t)world * nq * k;
    uint64_t* keys = nullptr;
    VQ_HIP(hipMallocAsync((void**)&keys, (size_t)count * 8, st));
    hipLaunchKernelGGL(pack_keys_kernel, dim3(cdiv(count, 256)), dim3(256), 0, st, (const int32_t*)d_all_ids, (const float*)d_all_dist,
                       count, (int64_t)0, keys);
    hipLaunchKernelGGL(merge_keys_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, st, keys, world, nq, k, (int32_t*)d_ids, (float*)d_dist);
    VQ_HIP(hipGetLastError());
    VQ_HIP(hipFreeAsync(keys, st));
    return 0;
}

// Search over a row-sharded matrix: this rank's index holds rows [row_offset, row_offset + size) of the global
// matrix.  Local exact top-k (same modes as vq_index_search_device) -> keys with global ids -> ONE all-gather of
// nq*k 8-byte keys per rank -> merge on every rank.  Runs on the index's stream.
int vq_index_search_sharded(vq_index* idx, vq_comm* c, const void* d_queries, int nq, int k, int mode, int64_t row_offset,
                            void* d_ids, void* d_dist) {
    VQ_TRY(require_init());
    VQ_CHECK(idx && c && nq >= 0 && k > 0 && (nq == 0 || (d_queries && d_ids && d_dist)), "vq_index_search_sharded: bad argument");
    VQ_CHECK((int64_t)c->world * k <= MERGE_MAX, "vq_index_search_sharded: world*k = %d exceeds %d", c->world * k, MERGE_MAX);
    VQ_CHECK(row_offset >= 0 && row_offset < ((int64_t)1 << 31), "vq_index_search_sharded: row_offset out of range");
    if (nq == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    const int64_t count = (int64_t)nq * k;
    VQ_TRY(grow(c->d_lids, c->lids_cap, count));
    VQ_TRY(grow(c->d_ldist, c->ldist_cap, count));
    VQ_TRY(grow(c->d_keys, c->keys_cap, count));
    VQ_TRY(grow(c->d_all, c->all_cap, count * c->world));
    hipStream_t st = nullptr; int64_t size = 0;
    VQ_TRY(index_search_local(idx, (const float*)d_queries, nq, k, mode, c->d_lids, c->d_ldist, &st, &size));
    VQ_CHECK(row_offset + size <= ((int64_t)1 << 31), "vq_index_search_sharded: global row ids exceed 2^31");
    hipLaunchKernelGGL(pack_keys_kernel, dim3(cdiv(count, 256)), dim3(256), 0, st, c->d_lids, c->d_ldist, count, row_offset, c->d_keys);
    VQ_HIP(hipGetLastError());
    VQ_NCCL(rccl()->AllGather(c->d_keys, c->d_all, (size_t)count, ncclUint64, c->comm, st));
    hipLaunchKernelGGL(merge_keys_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, st, c->d_all, c->world, nq, k, (int32_t*)d_ids, (float*)d_dist);
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

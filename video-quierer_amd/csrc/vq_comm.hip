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
hipStream_t index_stream(vq_index* x);

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
// `stride` = words per rank in `keys` (nq*k, or nq*k + 1 when every rank's block ends in a STATUS word: 0 = its local scan
// succeeded).  A non-zero status word of ANY rank voids the whole call on every rank: all slots come back empty (id -1,
// +inf) and the number of the first failed rank + 1 is left in *peer_err (host-visible; vq_comm_check reports it) — a peer of
// a rank whose local scan failed gets a visibly empty answer and an error, never a silently partial list, and never a hang.
constexpr int MERGE_MAX = 1024;
__global__ __launch_bounds__(256)
void merge_keys_kernel(const uint64_t* __restrict__ keys, int world, int nq, int k, int32_t* __restrict__ out_ids,
                       float* __restrict__ out_dist, int64_t stride, int32_t* __restrict__ peer_err) {
    __shared__ uint64_t cand[4][MERGE_MAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    const int n = world * k;
    if (stride > (int64_t)nq * k) {                              // status words present (workgroup-uniform)
        int bad = 0;
        for (int r = world - 1; r >= 0; --r) if (keys[(size_t)r * stride + (size_t)nq * k] != 0) bad = r + 1;
        if (bad) {
            if (q < nq) for (int j = lane; j < k; j += 64) { out_ids[(size_t)q * k + j] = -1; out_dist[(size_t)q * k + j] = __builtin_inff(); }
            if (blockIdx.x == 0 && threadIdx.x == 0 && peer_err) *peer_err = bad;
            return;
        }
    }
    if (q < nq)
        for (int i = lane; i < n; i += 64) {
            const int r = i / k, j = i - r * k;
            cand[w][i] = keys[(size_t)r * stride + (size_t)q * k + j];
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

// [W][max rows][dim] padded gather -> the ranks' valid prefixes back to back (rank = frame order)
int compact_gathered(const float* d_padded, const int64_t* counts, int world, int64_t pad_rows, int dim, float* d_out, hipStream_t st) {
    int64_t off = 0;
    for (int r = 0; r < world; ++r) {
        if (counts[r] > 0)
            VQ_HIP(hipMemcpyAsync(d_out + off * dim, d_padded + (int64_t)r * pad_rows * dim, (size_t)counts[r] * dim * 4,
                                  hipMemcpyDeviceToDevice, st));
        off += counts[r];
    }
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
    // What all ranks have AGREED the scratch holds (words per rank of the key exchange; floats per rank of the padded
    // gather).  Every rank walks the same sequence of calls, so every rank sees a call outgrow these at the same call — and
    // only then do they exchange one status word each (agree()) before anybody enters the data collective.
    int64_t agreed_keys = 0, agreed_pad = 0;
    uint64_t* d_status = nullptr;      // [world + 1]: gathered status words, then mine (made at init: agree() allocates nothing)
    uint64_t* h_status = nullptr;      // pinned mirror
    int32_t* h_peer_err = nullptr;     // pinned + mapped: merge_keys_kernel leaves (first failed rank + 1) here
    int32_t* d_peer_err = nullptr;     // the device's address of the same word
    std::vector<void*> retired;        // outgrown scratch: freed at destroy (hipFree inside a stream-ordered path waits for the whole device)
    int fail_next_alloc = 0;           // $VQ_COMM_FAIL_ALLOC (tests): that many scratch allocations fail
};

namespace {

template <class T> int grow(vq_comm* c, T*& p, int64_t& cap, int64_t need) {
    if (need <= cap) return 0;
    const int64_t ncap = std::max<int64_t>(need, cap * 2);
    T* np = nullptr;
    hipError_t e = c->fail_next_alloc > 0 ? (--c->fail_next_alloc, hipErrorOutOfMemory) : hipMalloc((void**)&np, (size_t)ncap * sizeof(T));
    if (e != hipSuccess) return fail(VQ_ERR_OOM, "vq_comm[rank %d]: scratch hipMalloc(%lld) failed: %s", c->rank, (long long)(ncap * sizeof(T)), hipGetErrorString(e));
    if (p) c->retired.push_back(p);               // work already enqueued may still read it
    p = np; cap = ncap;
    return 0;
}

// One status word per rank, gathered and READ (the stream is waited for): 0 everywhere, or every rank returns the same error.
// Called only where the ranks would otherwise part ways — when a call outgrows the agreed scratch (an allocation may fail
// on one rank only) — never on the steady-state path.  Uses buffers made at vq_comm_init.
int agree(vq_comm* c, int local_rc, hipStream_t st, const char* what) {
    const std::string local_msg = local_rc ? last_error() : std::string();
    c->h_status[c->world] = (uint64_t)(uint32_t)(local_rc ? -local_rc : 0);
    VQ_HIP(hipMemcpyAsync(c->d_status + c->world, c->h_status + c->world, 8, hipMemcpyHostToDevice, st));
    VQ_NCCL(rccl()->AllGather(c->d_status + c->world, c->d_status, 1, ncclUint64, c->comm, st));
    VQ_HIP(hipMemcpyAsync(c->h_status, c->d_status, (size_t)c->world * 8, hipMemcpyDeviceToHost, st));
    VQ_HIP(hipStreamSynchronize(st));
    for (int r = 0; r < c->world; ++r)
        if (c->h_status[r] != 0) {
            if (r == c->rank) return fail(local_rc, "%s", local_msg.c_str());
            return fail(VQ_ERR_STATE, "vq_comm[rank %d]: rank %d could not prepare %s (its error code %d): no rank entered the exchange",
                        c->rank, r, what, -(int)c->h_status[r]);
        }
    return 0;
}

}  // namespace

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
    // the status exchange's buffers exist from here on: agree() must not need an allocation that could itself fail
    hipError_t e = hipMalloc((void**)&c->d_status, (size_t)(world + 1) * 8);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_status, (size_t)(world + 1) * 8);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_peer_err, 4, hipHostMallocMapped);
    if (e == hipSuccess) { *c->h_peer_err = 0; e = hipHostGetDevicePointer((void**)&c->d_peer_err, c->h_peer_err, 0); }
    if (e != hipSuccess) { vq_comm_destroy(c); return fail(VQ_ERR_OOM, "vq_comm_init: status buffers: %s", hipGetErrorString(e)); }
    if (const char* fa = getenv("VQ_COMM_FAIL_ALLOC")) c->fail_next_alloc = atoi(fa);
    *out = c;
    return 0;
}

int vq_comm_destroy(vq_comm* c) {
    if (!c) return 0;
    (void)hipDeviceSynchronize();
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    (void)hipFree(c->d_lids); (void)hipFree(c->d_ldist); (void)hipFree(c->d_keys); (void)hipFree(c->d_all);
    (void)hipFree(c->d_pad); (void)hipFree(c->d_gath); (void)hipFree(c->d_status);
    for (void* p : c->retired) (void)hipFree(p);
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->h_peer_err) (void)hipHostFree(c->h_peer_err);
    delete c;
    return 0;
}

// Has a call on this communicator been voided by a PEER's failure?  vq_index_search_sharded is asynchronous: a rank whose own
// scan fails returns that error at once, but still enters the exchange with a non-zero status word; its peers learn of it on
// the device, hand back empty lists (every id -1) and leave the failed rank's number here.  Call after synchronising the
// stream; reports once, then clears.
int vq_comm_check(vq_comm* c) {
    VQ_CHECK(c, "vq_comm_check: null handle");
    std::lock_guard<std::mutex> lk(c->mu);
    const int bad = *(volatile int32_t*)c->h_peer_err;
    if (bad == 0) return 0;
    *c->h_peer_err = 0;
    return fail(VQ_ERR_STATE, "vq_comm[rank %d]: a sharded search was voided: the local scan of rank %d failed (every rank's result "
                "lists of that call are empty)", c->rank, bad - 1);
}

// The second half of a ragged vq_allgather_rows on its own: [world][pad_rows][dim] fp32 (every rank's rows padded to pad_rows)
// -> the valid prefixes counts[r] back to back at d_out.  Device pointers; asynchronous on hip_stream.
int vq_compact_gathered_rows(const void* d_padded, const int64_t* counts, int world, int64_t pad_rows, int dim, void* d_out, void* hip_stream) {
    VQ_TRY(require_init());
    VQ_CHECK(d_padded && counts && world >= 1 && pad_rows >= 0 && dim > 0 && d_out, "vq_compact_gathered_rows: bad argument");
    for (int r = 0; r < world; ++r) VQ_CHECK(counts[r] >= 0 && counts[r] <= pad_rows, "vq_compact_gathered_rows: counts[%d] = %lld outside [0, %lld]", r, (long long)counts[r], (long long)pad_rows);
    return compact_gathered((const float*)d_padded, counts, world, pad_rows, dim, (float*)d_out, (hipStream_t)hip_stream);
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
    // Argument checks every rank fails alike (same counts / dim on every rank by contract) come first: nobody has entered
    // anything yet.
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
    // Ragged: pad to the largest count, gather, compact.  A call that outgrows the agreed scratch allocates first and then
    // exchanges one status word per rank — a rank whose hipMalloc failed reports it THERE, and every rank returns an error
    // without having entered the data collective (its peers would have waited in it for ever).
    if (mx * dim > c->agreed_pad) {
        int rc = grow(c, c->d_pad, c->pad_cap, mx * dim);
        if (rc == 0) rc = grow(c, c->d_gath, c->gath_cap, mx * dim * c->world);
        VQ_TRY(agree(c, rc, st, "the padded all-gather's scratch"));
        c->agreed_pad = mx * dim;
    }
    VQ_HIP(hipMemsetAsync(c->d_pad, 0, (size_t)mx * dim * 4, st));
    if (counts[c->rank] > 0)
        VQ_HIP(hipMemcpyAsync(c->d_pad, d_local, (size_t)counts[c->rank] * dim * 4, hipMemcpyDeviceToDevice, st));
    VQ_NCCL(r->AllGather(c->d_pad, c->d_gath, (size_t)mx * dim, ncclFloat, c->comm, st));
    return compact_gathered(c->d_gath, counts, c->world, mx, dim, (float*)d_out, st);
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
    hipLaunchKernelGGL(merge_keys_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, st, keys, world, nq, k, (int32_t*)d_ids, (float*)d_dist,
                       (int64_t)nq * k, (int32_t*)nullptr);
    VQ_HIP(hipGetLastError());
    VQ_HIP(hipFreeAsync(keys, st));
    return 0;
}

// Search over a row-sharded matrix: this rank's index holds rows [row_offset, row_offset + size) of the global
// matrix.  Local exact top-k (same modes as vq_index_search_device) -> keys with global ids -> ONE all-gather of
// nq*k 8-byte keys (+ one status word) per rank -> merge on every rank.  Runs on the index's stream.
//
// No rank may be left waiting in the collective for a peer that returned early:
//   1. checks every rank fails alike (nq, k, world * k, null pointers) come first — nobody has entered anything;
//   2. a call that outgrows the agreed scratch allocates and then exchanges one status word per rank (agree()): an
//      allocation that failed on one rank makes EVERY rank return an error, before the data collective;
//   3. whatever can fail on one rank only after that — its local scan (a shard the fp16 path refuses, a stale id-rank table,
//      a launch error), a row_offset its shard overflows — does not stop that rank from entering the exchange: it sends
//      empty keys and a non-zero status word, then returns its error.  The peers' merge kernel sees the word, hands back
//      empty lists and flags the communicator (vq_comm_check).
int vq_index_search_sharded(vq_index* idx, vq_comm* c, const void* d_queries, int nq, int k, int mode, int64_t row_offset,
                            void* d_ids, void* d_dist) {
    VQ_TRY(require_init());
    VQ_CHECK(idx && c && nq >= 0 && k > 0 && (nq == 0 || (d_queries && d_ids && d_dist)), "vq_index_search_sharded: bad argument");
    VQ_CHECK((int64_t)c->world * k <= MERGE_MAX, "vq_index_search_sharded: world*k = %d exceeds %d", c->world * k, MERGE_MAX);
    if (nq == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    const int64_t count = (int64_t)nq * k, words = count + 1;
    hipStream_t st = index_stream(idx);
    if (words > c->agreed_keys) {
        int rc = grow(c, c->d_lids, c->lids_cap, count);
        if (rc == 0) rc = grow(c, c->d_ldist, c->ldist_cap, count);
        if (rc == 0) rc = grow(c, c->d_keys, c->keys_cap, words);
        if (rc == 0) rc = grow(c, c->d_all, c->all_cap, words * c->world);
        VQ_TRY(agree(c, rc, st, "the sharded search's scratch"));
        c->agreed_keys = words;
    }
    int64_t size = 0;
    int local_rc = 0;
    std::string local_msg;
    if (!(row_offset >= 0 && row_offset < ((int64_t)1 << 31)))
        local_rc = fail(VQ_ERR_INVALID, "vq_index_search_sharded: row_offset out of range");
    if (local_rc == 0) {
        hipStream_t st2 = nullptr;
        local_rc = index_search_local(idx, (const float*)d_queries, nq, k, mode, c->d_lids, c->d_ldist, &st2, &size);
    }
    if (local_rc == 0 && row_offset + size > ((int64_t)1 << 31))
        local_rc = fail(VQ_ERR_INVALID, "vq_index_search_sharded: global row ids exceed 2^31");
    if (local_rc != 0) local_msg = last_error();
    // From here to the collective nothing returns: errors are folded into the status word.
    hipError_t he = hipSuccess;
    if (local_rc == 0) {
        hipLaunchKernelGGL(pack_keys_kernel, dim3(cdiv(count, 256)), dim3(256), 0, st, c->d_lids, c->d_ldist, count, row_offset, c->d_keys);
        he = hipGetLastError();
        if (he == hipSuccess) he = hipMemsetAsync(c->d_keys + count, 0, 8, st);
        if (he != hipSuccess) { local_rc = VQ_ERR_HIP; local_msg = std::string("vq_index_search_sharded: packing the keys failed: ") + hipGetErrorString(he); }
    }
    if (local_rc != 0) (void)hipMemsetAsync(c->d_keys, 0xFF, (size_t)words * 8, st);      // every slot empty, status word non-zero
    const ncclResult_t nr = rccl()->AllGather(c->d_keys, c->d_all, (size_t)words, ncclUint64, c->comm, st);
    if (nr != ncclSuccess) return fail(VQ_ERR_HIP, "vq_index_search_sharded: ncclAllGather failed: %s", rccl()->GetErrorString(nr));
    hipLaunchKernelGGL(merge_keys_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, st, c->d_all, c->world, nq, k, (int32_t*)d_ids, (float*)d_dist,
                       words, c->d_peer_err);
    if (local_rc != 0) return fail(local_rc, "%s", local_msg.c_str());
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

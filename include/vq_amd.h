/* libvq_amd — C ABI of the MI355X (gfx950) frame-embedding + exact cosine k-NN path.
 *
 * The reference (adhney/video-quierer) has no FFI boundary: its boundary for this
 * path is two Python classes,
 *   FeatureExtractor        reference src/core/feature_extractor.py:21-258
 *   HNSWIndex / OptimizedHNSWIndex   reference src/indexes/hnsw.py:19-528
 * constructed at reference src/video_search_system.py:65-69 and :79-89.  The
 * build's same-named Python classes (video-quierer_amd/core/feature_extractor.py,
 * video-quierer_amd/indexes/hnsw.py) bind exactly the entry points below with
 * ctypes; INTEGRATION.md shows the binding.  Each group names the reference
 * method(s) it replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a negative code; the message is in
 *     vq_last_error() (thread-local).  Wrappers raise (reference convention:
 *     log + raise, feature_extractor.py:175-177, hnsw.py:353-354).
 *   - the caller owns every host buffer (C-contiguous); the library owns device
 *     memory behind the opaque handles; *_destroy frees it.
 *   - handles are safe to share between threads (one mutex + one HIP stream per
 *     handle); ctypes releases the GIL during calls.
 *   - "_device" variants take device pointers (e.g. torch.Tensor.data_ptr()) and
 *     are asynchronous on the handle's stream until *_synchronize.
 *   - no torch / STL types cross this boundary.
 */
#ifndef VQ_AMD_H
#define VQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQ_OK 0
#define VQ_ERR_INVALID (-1)
#define VQ_ERR_HIP (-2)
#define VQ_ERR_STATE (-3)
#define VQ_ERR_OOM (-4)

/* ---- library ------------------------------------------------------------- */
/* Binds the calling process to one GPU (reference: device pick in
 * FeatureExtractor.__init__, feature_extractor.py:41-44).  Fails when no gfx950
 * device is visible: there is no CPU fallback. */
int vq_init(int device_ordinal);
int vq_device_count(int* count);
const char* vq_last_error(void);
const char* vq_version(void);

/* ---- encoder: FeatureExtractor._load_model / extract_batch ----------------- */
typedef struct vq_encoder vq_encoder;

typedef struct vq_vit_config {
    int32_t image_size;   /* 224 */
    int32_t patch_size;   /* 32  */
    int32_t hidden;       /* 768 */
    int32_t mlp;          /* 3072 */
    int32_t layers;       /* 12 */
    int32_t heads;        /* 12 */
    int32_t proj_dim;     /* 512 */
    float ln_eps;         /* 1e-5 */
} vq_vit_config;

/* weights: n_weights host fp32 tensors in the order of
 * video-quierer_amd/weights.py:weight_shapes() (HF state_dict names; 5 + 16*layers + 3).
 * Replaces CLIPModel.from_pretrained + .to(device) (feature_extractor.py:76-81).
 * max_batch: frames per device pass (workspace is sized for it). */
int vq_encoder_create(const vq_vit_config* cfg, const float* const* weights, int n_weights,
                      int max_batch, vq_encoder** out);
/* Same with flags: VQ_ENC_FP16 = fp16 instead of bf16 GEMM operands in every group (same MFMA rate, ~8x smaller
 * rounding error; what the Python host passes by default and the type ViT-L/14@336 is specified with; the patch
 * weights W/(255 std) are scaled by a power of two out of fp16's subnormal range and the scale is undone, exactly,
 * in the GEMM epilogue).  $VQ_AMD_DTYPE=fp16|bf16 overrides. */
#define VQ_ENC_FP16 1
/* Operand type per GEMM group (set = fp16, clear = bf16; fp32 accumulation and the same MFMA rate either way):
 * PATCH = pixels x W_patch, QKV = LN1 output x W_qkv, ATTN = q|k|v, softmax P, attention output x W_out,
 * FC1 = LN2 output x W_fc1, FC2 = quick-GELU output x W_fc2.  VQ_ENC_MIXED (all but the patch GEMM) was round 2's
 * default (which roundings the 1e-3 score tolerance can afford).  $VQ_AMD_DTYPE =
 * bf16 | fp16 | mixed | mask:<bits 0-4> overrides the flags. */
#define VQ_ENC_F16_PATCH 0x100
#define VQ_ENC_F16_QKV 0x200
#define VQ_ENC_F16_ATTN 0x400
#define VQ_ENC_F16_FC1 0x800
#define VQ_ENC_F16_FC2 0x1000
#define VQ_ENC_MIXED (VQ_ENC_F16_QKV | VQ_ENC_F16_ATTN | VQ_ENC_F16_FC1 | VQ_ENC_F16_FC2)
/* VQ_ENC_CONCURRENT: the caller keeps several encoder handles busy at once on separate HIP streams.  The
 * N = hidden GEMMs then keep 256-row tiles (150 dense workgroups at batch 256, leaving CUs to the other
 * streams: +7 % aggregate frames/s measured with 3 streams) instead of the 160-row tiles that spread one
 * pass over 240 CUs (+5 % for a single stream). */
#define VQ_ENC_CONCURRENT 2
int vq_encoder_create_ex(const vq_vit_config* cfg, const float* const* weights, int n_weights,
                         int max_batch, int flags, vq_encoder** out);
/* Another handle on the SAME device weights (own stream, workspace and profiling state): for callers that keep
 * several batches in flight.  flags: VQ_ENC_CONCURRENT only (operand types are the parent's).  The weights live
 * until the last handle using them is destroyed, in any order. */
int vq_encoder_create_shared(vq_encoder* parent, int max_batch, int flags, vq_encoder** out);
int vq_encoder_destroy(vq_encoder* enc);

/* extract_batch (feature_extractor.py:137-177) for uint8 frames already at
 * image_size x image_size: frames [n][S][S][3] -> out [n][proj_dim] fp32,
 * L2-normalised.  swap_rb=1 is the ndarray path (BGR->RGB, :111-112), 0 the PIL
 * path.  Any n >= 0 (processed in max_batch slices); synchronous. */
int vq_encoder_encode_u8(vq_encoder* enc, const uint8_t* frames, int n, int swap_rb, float* out);

/* Same, device-resident input/output, n <= max_batch, asynchronous.
 * d_out_f16 (optional, may be NULL) receives an fp16 copy of the embeddings. */
int vq_encoder_encode_u8_device(vq_encoder* enc, const void* d_frames, int n, int swap_rb,
                                void* d_out_f32, void* d_out_f16);
/* Pinned host staging: two slots of max_batch frames each.  The host side assembles frames straight
 * into a slot (one copy instead of two) and encodes from it; while slot s is being encoded (the call
 * blocks only its own thread) another thread may fill slot 1-s.  n <= max_batch. */
int vq_encoder_staging(vq_encoder* enc, int slot, uint8_t** host_ptr, size_t* bytes);
int vq_encoder_encode_staged(vq_encoder* enc, int slot, int n, int swap_rb, float* out);
/* Pipelined ingest of host frames (what extract_from_video_frames, feature_extractor.py:179-209, loops over):
 *   vq_encoder_stage_frames   gathers n separately allocated S x S x 3 uint8 frames (a Python list of ndarray
 *                             frames) into pinned slot 0/1 on up to n_threads host threads;
 *   vq_encoder_submit_staged  enqueues upload (copy stream) -> forward -> download for that slot and returns;
 *   vq_encoder_wait_staged    blocks until the slot's batch is done and copies out [n][proj_dim] fp32.
 * With two slots the upload and the host gather of batch i+1 overlap the forward pass of batch i. */
int vq_encoder_stage_frames(vq_encoder* enc, int slot, const uint8_t* const* frames, int n, int n_threads);
int vq_encoder_submit_staged(vq_encoder* enc, int slot, int n, int swap_rb);
int vq_encoder_wait_staged(vq_encoder* enc, int slot, float* out);
int vq_encoder_synchronize(vq_encoder* enc);
/* Run this handle's kernels on a caller-owned HIP stream (e.g. torch's current
 * stream, so RCCL collectives issued by torch order after the encode without a
 * host sync).  NULL restores the handle's own stream. */
int vq_encoder_set_stream(vq_encoder* enc, void* hip_stream);
int vq_encoder_output_dim(vq_encoder* enc, int* dim);

/* Per-kernel-class device timing with HIP events on the encoder's stream
 * (bench.py roofline leg).  Between begin/end every launch is bracketed by
 * events; end() reports total ms and launch count per class. */
#define VQ_ENC_NCLASS 11
int vq_encoder_profile_begin(vq_encoder* enc);
int vq_encoder_profile_end(vq_encoder* enc, float* ms /*[VQ_ENC_NCLASS]*/, int* launches /*[VQ_ENC_NCLASS]*/);
const char* vq_encoder_profile_class_name(int cls);
/* Median elapsed time of an EMPTY event bracket on the encoder's stream: what a bracket adds to the kernel it holds. */
int vq_encoder_profile_bracket_overhead(vq_encoder* enc, float* ms);

/* Test hooks: run only the first `layers` transformer blocks (<0: all), and copy
 * an internal activation to the host as fp32: "x" [n*T][hidden] residual stream,
 * "h" LN output, "qkv", "att", "mlp" (bf16 widened). */
int vq_encoder_debug_set_layers(vq_encoder* enc, int layers);
int vq_encoder_debug_read(vq_encoder* enc, const char* name, int rows, float* out);

/* C[M][N] = A[M][K] * W[N][K]^T through the production MFMA mainloops (inputs
 * rounded from the given fp32, fp32 accumulate) — unit-test hook.
 * flags: bit 0 = fp16 inputs (else bf16); bits 1-4 = kernel (0 auto, 1 = 128x128
 * two-phase, 2 = 256x256 phased). */
int vq_debug_gemm(const float* A, const float* W, int M, int N, int K, int flags, float* C);

/* ---- text tower: FeatureExtractor.extract_text_features (feature_extractor.py:218-234) ------------ */
/* CLIPTextModel + text_projection on token ids (the tokenizer stays on the host).  weights: host fp32
 * tensors in the order of video-quierer_amd/weights.py:text_weight_shapes() (2 + 16*layers + 3).
 * flags as vq_encoder_create_ex. */
typedef struct vq_encoder vq_text_encoder;
typedef struct vq_text_config {
    int32_t vocab;          /* 49408 */
    int32_t max_positions;  /* 77 */
    int32_t hidden;         /* 512 */
    int32_t mlp;            /* 2048 */
    int32_t layers;         /* 12 */
    int32_t heads;          /* 8 */
    int32_t proj_dim;       /* 512 */
    int32_t eos_token_id;   /* 49407 */
    float ln_eps;           /* 1e-5 */
} vq_text_config;
int vq_text_encoder_create(const vq_text_config* cfg, const float* const* weights, int n_weights,
                           int max_batch, int flags, vq_text_encoder** out);
/* ids [n][seq_len] (seq_len <= max_positions; every row holds an eos token, as the tokenizer produces):
 * out [n][proj_dim] fp32, L2-normalised.  Rows are padded to max_positions with eos internally (the
 * causal mask makes the padding invisible to the pooled EOS position).  Synchronous; any n. */
int vq_text_encoder_encode_ids(vq_text_encoder* enc, const int32_t* ids, int n, int seq_len, float* out);
int vq_text_encoder_destroy(vq_text_encoder* enc);

/* Mainloop stamps / ablations / clock probes are not product entry points: include/vq_amd_diag.h (`make DIAG=1`). */

/* ---- index: HNSWIndex.add / search / size / save / load --------------------- */
typedef struct vq_index vq_index;

/* HNSWIndex.__init__ (hnsw.py:25-57).  The graph parameters (M, ef_*) have no
 * device-side meaning: the index is an exact scan. */
int vq_index_create(int dim, vq_index** out);
int vq_index_destroy(vq_index* idx);

/* add / add_batch (hnsw.py:150-236): appends n rows.  normalize=1 divides each
 * row by its L2 norm on the device (fixed-order fp64 chain, see oracle/knn_oracle.c);
 * normalize=0 stores the rows as given (the Python wrapper normalises with numpy
 * exactly like the reference, hnsw.py:157, and passes 0).  Rows stored as given are MEASURED, not trusted:
 * the index keeps the range of |row|^2 it holds; while 0.5 <= |row|^2 <= 2 the fp16 scan stays available
 * with its error bound scaled by the largest |row|, outside that range mode 0 uses the exact scan and
 * mode 2 is refused. */
int vq_index_add(vq_index* idx, const float* rows, int64_t n, int normalize);
int vq_index_add_device(vq_index* idx, const void* d_rows_f32, int64_t n, int normalize);
/* Re-add of an id the index already holds (hnsw.py:160 `self.data[node_id] = vector`: a dict assignment): replaces
 * stored rows IN PLACE — fp32 master, fp16 scan copy and the |row|^2 range — without touching the other rows.
 * rows [n][dim] (host), row_numbers [n] in [0, size).  A row named more than once keeps its LAST update, as the
 * reference's sequential assignments do.  normalize as vq_index_add.  The result is bit-identical to an index built
 * from scratch with the updated rows.  Synchronous. */
int vq_index_update_rows(vq_index* idx, const float* rows, const int64_t* row_numbers, int64_t n, int normalize);
int vq_index_size(vq_index* idx, int64_t* n);
int vq_index_clear(vq_index* idx);
/* The tie order of the result lists.  The reference returns `sorted(candidates)[:k]` over (distance, id) tuples
 * (hnsw.py:269, :518): rows at EQUAL distance (duplicate frames) come back in the order of the caller's ids — strings
 * f"{video_id}_{i}" under video_search_system.py:164-166, where "video0_10" sorts before "video0_2".  The library sees row
 * numbers only, so the host hands it rank_of_row[r] = position of row r's id in the caller's id order (a permutation of
 * 0..n-1, n = the current size; checked).  Every search then selects and orders by (dist, rank) on the device and still
 * reports ROW numbers.  n = 0 (rank_of_row may be NULL) returns to (dist, row) order.  Ranks describe the rows present when
 * they were set: after vq_index_add* a search is refused (VQ_ERR_INVALID) until the ranks are set again or cleared;
 * vq_index_update_rows keeps them (same ids), vq_index_clear drops them.  Synchronous.  vq_index_search_sharded orders
 * ties ACROSS shards by global row number whatever the shards' ranks say. */
int vq_index_set_id_ranks(vq_index* idx, const int32_t* rank_of_row, int64_t n);

/* search / search_batch (hnsw.py:238-300, 488-528) as an exact scan:
 *   dist = fp32(1 - fp32(dot(row, q))), k smallest, ordered by (dist, row) — or (dist, id rank) once
 *   vq_index_set_id_ranks has been called.
 * queries [nq][dim] are used as given (the wrapper does query / ||query|| with numpy, hnsw.py:250).  The
 * fp16 path's exactness bound scales with each query's own norm while 0.25 <= |q|^2 <= 4; a query outside that
 * range (or not finite) is outside what fp16 operands can bound and is answered by the exact scan instead
 * (device-side fallback), so un-normalised queries stay exact at any scale.  ids/dist are [nq][k]; unused slots
 * (k > size) are id -1 / dist +inf.  mode: 0 auto (fp16 scan from 16,384 rows and k <= 100: the API takes k up to 50, src/api/routes.py:58, and the caller searches for k * 2, video_search_system.py:297), 1 exact
 * fp32-master scan, 2 fp16 MFMA scan + exact re-score with proof (unproven queries are redone by the
 * exact scan, on the device: nothing is read back).  vq_index_search_device is asynchronous on the index's
 * stream in every mode, with one exception: the first search after a vq_index_add_device(normalize=0) waits
 * for the stream once to read the |row|^2 range those rows were measured at (the host-side adds read it
 * before they return).  vq_index_last_search_stats waits for that stream. */
int vq_index_search(vq_index* idx, const float* queries, int nq, int k, int mode,
                    int32_t* ids, float* dist);
int vq_index_search_device(vq_index* idx, const void* d_queries_f32, int nq, int k, int mode,
                           void* d_ids_i32, void* d_dist_f32);
int vq_index_synchronize(vq_index* idx);
int vq_index_set_stream(vq_index* idx, void* hip_stream);

/* save / load support (hnsw.py:306-380): the stored (normalised) rows. */
int vq_index_export(vq_index* idx, float* rows /*[size][dim]*/);
/* Single stored rows (the reference reads `self.data[node_id]`, a dict lookup): out [n][dim] = rows row_numbers[0..n). */
int vq_index_read_rows(vq_index* idx, const int64_t* row_numbers, int64_t n, float* out);

/* ---- multi-GPU exchange over RCCL (xGMI): one process per GPU ------------------------------------
 * The reference is single-device; these entry points are what a multi-GPU deployment of its
 * ingest loop (src/video_search_system.py:152-181) and of its search (:297) adds: an all-gather of the per-shard
 * embeddings before indexing, and for a row-sharded matrix an all-gather of the per-shard top-k followed by a
 * k-way merge in the reference's (distance, id) order (src/indexes/hnsw.py:269).  librccl is loaded on first
 * use.  Rank 0 makes the id (vq_comm_unique_id, 128 bytes) and ships it to the other ranks by any side
 * channel (torch.distributed's store, a file, MPI); every rank then calls vq_comm_init after vq_init. */
typedef struct vq_comm vq_comm;
#define VQ_COMM_ID_BYTES 128
int vq_comm_unique_id(void* out_id, int bytes);
int vq_comm_init(int rank, int world, const void* unique_id, vq_comm** out);
int vq_comm_destroy(vq_comm* comm);
int vq_comm_info(vq_comm* comm, int* rank, int* world, int* rccl_version);
/* Collectives and failures.  Every rank must make the same sequence of calls with the same counts / nq / k.  No call leaves a
 * rank waiting in a collective its peer never enters: argument checks that fail alike on every rank come first; a call
 * that needs more scratch than the ranks have agreed on allocates and then exchanges ONE status word per rank (an
 * allocation that failed anywhere makes every rank return an error before the data collective; steady-state calls skip
 * this); a failure that can strike one rank only after that point is carried INTO the collective as a status word (below). */
/* counts[world]: rows each rank contributes (this rank's d_local holds counts[rank] x dim fp32); d_out receives
 * sum(counts) x dim in rank (= frame) order on every rank.  Asynchronous on hip_stream, except for the call that first
 * needs (more) padding scratch for ragged counts: that one waits for the stream once (status exchange). */
int vq_allgather_rows(vq_comm* comm, const void* d_local, const int64_t* counts, int dim, void* d_out, void* hip_stream);
/* The ragged all-gather's second half on its own: d_padded [world][pad_rows][dim] fp32 -> the first counts[r] rows of every
 * rank's block, back to back, at d_out.  Asynchronous on hip_stream. */
int vq_compact_gathered_rows(const void* d_padded, const int64_t* counts, int world, int64_t pad_rows, int dim, void* d_out,
                             void* hip_stream);
/* This rank's index holds rows [row_offset, row_offset + size) of the global matrix.  Same modes and result
 * layout as vq_index_search_device, ids are GLOBAL row numbers, identical on every rank.  world*k <= 1024.
 * A rank whose LOCAL scan fails (its shard refuses the requested mode, stale id ranks, a launch error, a row_offset its
 * shard overflows) returns that error — after entering the exchange with empty keys and a non-zero status word.  On its
 * peers the call itself returns 0 (it is asynchronous), every result slot comes back empty (id -1, +inf) and
 * vq_comm_check reports the failed rank once the stream has been synchronised.  Ties across shards are ordered by global
 * row number (vq_index_set_id_ranks orders ties inside a shard only). */
int vq_index_search_sharded(vq_index* idx, vq_comm* comm, const void* d_queries_f32, int nq, int k, int mode,
                            int64_t row_offset, void* d_ids_i32, void* d_dist_f32);
/* 0, or VQ_ERR_STATE once after a sharded search on this communicator was voided by a peer's local failure. */
int vq_comm_check(vq_comm* comm);
/* The merge step alone: [world][nq][k] shard results with global ids (-1 / +inf = empty slot) -> [nq][k]. */
int vq_merge_topk_device(const void* d_all_ids_i32, const void* d_all_dist_f32, int world, int nq, int k,
                         void* d_ids_i32, void* d_dist_f32, void* hip_stream);

/* Device timing of the scan kernels between begin/end (bench.py roofline leg). */
#define VQ_IDX_NCLASS 6
int vq_index_profile_begin(vq_index* idx);
int vq_index_profile_end(vq_index* idx, float* ms /*[VQ_IDX_NCLASS]*/, int* launches /*[VQ_IDX_NCLASS]*/);
const char* vq_index_profile_class_name(int cls);
/* counters of the last fp16-scan search: [0] queries verified exact by the
 * margin test, [1] queries that needed block rescans, [2] queries sent to the
 * full exact scan */
int vq_index_last_search_stats(vq_index* idx, int64_t* stats /*[3]*/);

/* ------------------------------------------------------------------ frame preprocessing
 * The resize in front of the encoder, bit-identical to Pillow's 8-bit separable resample
 * (Pillow src/libImaging/Resample.c), which is what the reference runs in two places:
 *   - transforms.Resize((224, 224)) on a PIL image, feature_extractor.py:54-61, :105-116
 *       -> filter BILINEAR, out 224x224, crop = the whole resized frame;
 *   - CLIPProcessor on the live path, video_search_overhaul.py:129-135, :218-221, :283-289
 *       -> filter BICUBIC, short edge to 224, centre crop 224x224 (vq_clip_processor_geometry).
 * Frames are uint8 [n][h][w][3] of one size; channel order is untouched (resize is per channel).
 * The resized frame is out_h x out_w; only the window crop_h x crop_w at (crop_top, crop_left) is produced:
 * out [n][crop_h][crop_w][3]. */
#define VQ_RESAMPLE_BILINEAR 2      /* PIL.Image.BILINEAR */
#define VQ_RESAMPLE_BICUBIC 3       /* PIL.Image.BICUBIC */
/* cv2.resize(frame, (out_w, out_h)) with the default INTER_LINEAR — what OptimizedFrameExtractor applies
 * (frame_extractor.py:283-284): OpenCV's 11-bit two-tap bilinear without antialiasing, its exact-2x shortcut
 * to the 2x2 area average, equal sizes copied.  OpenCV is absent from the build container: restated from
 * modules/imgproc/src/resize.cpp, parity unpinned. */
#define VQ_RESAMPLE_CV_LINEAR 100
typedef struct vq_resampler vq_resampler;
int vq_resampler_create(vq_resampler** out);
int vq_resampler_destroy(vq_resampler* r);
int vq_resampler_set_stream(vq_resampler* r, void* hip_stream);
int vq_resampler_synchronize(vq_resampler* r);
/* Host frames in; host result out, or (out == NULL) left in the handle's device buffer for
 * vq_encoder_encode_u8_device.  Returns when the result is complete. */
int vq_resampler_run_u8(vq_resampler* r, const uint8_t* frames, int n, int h, int w, int filter,
                        int out_h, int out_w, int crop_top, int crop_left, int crop_h, int crop_w, uint8_t* out);
/* Same for n separately allocated frames of one size (a Python list of ndarray frames). */
int vq_resampler_run_u8_list(vq_resampler* r, const uint8_t* const* frames, int n, int h, int w, int filter,
                             int out_h, int out_w, int crop_top, int crop_left, int crop_h, int crop_w, uint8_t* out);
/* Device frames in, device result out (d_out == NULL: the handle's buffer); asynchronous on the handle's stream. */
int vq_resampler_run_u8_device(vq_resampler* r, const uint8_t* d_frames, int n, int h, int w, int filter,
                               int out_h, int out_w, int crop_top, int crop_left, int crop_h, int crop_w, uint8_t* d_out);
int vq_resampler_device_output(vq_resampler* r, void** d_ptr, int64_t* bytes);
/* Output size and crop offsets of the CLIP image processor for an h x w frame
 * (transformers image_transforms.py:295-299 get_resize_output_image_size + centre crop). */
int vq_clip_processor_geometry(int h, int w, int size, int crop, int* resized_h, int* resized_w,
                               int* crop_top, int* crop_left);
/* OptimizedFrameExtractor._is_low_quality inputs (frame_extractor.py:301-316) for BGR uint8 frames:
 * mean_brightness[i] = np.mean(frame_i); laplacian_var[i] = cv2.Laplacian(BGR2GRAY(frame_i), CV_64F).var().
 * on_device != 0: `frames` is a device pointer.  (OpenCV is absent from the build container: this
 * restates its published fixed-point grey conversion and 4-neighbour stencil; parity unpinned.) */
int vq_frame_quality_u8(vq_resampler* r, const uint8_t* frames, int n, int h, int w, int on_device,
                        double* mean_brightness, double* laplacian_var);

#ifdef __cplusplus
}
#endif
#endif /* VQ_AMD_H */

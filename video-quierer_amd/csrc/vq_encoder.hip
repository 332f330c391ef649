// vq_encoder: CLIP ViT image-encoder forward pass on one MI355X.
// Replaces FeatureExtractor._load_model / extract_batch
// (reference src/core/feature_extractor.py:70-103, 137-177) and the
// transformers CLIP vision tower it calls.
#include "../../include/vq_amd.h"
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "gemm_mfma256p.h"
#include "encoder_kernels.h"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <type_traits>
#include <thread>
#include <algorithm>
#include <vector>

namespace vq {
int require_init();

enum EncClass { C_PATCHIFY = 0, C_GEMM_PATCH, C_EMBED_FINISH, C_LAYERNORM, C_GEMM_QKV, C_ATTENTION,
                C_GEMM_OUT, C_GEMM_FC1, C_GEMM_FC2, C_POOL, C_LAST_CLS };
static const char* kEncClassNames[VQ_ENC_NCLASS] = {
    "patchify_u8", "gemm_patch_embed", "embed_finish_ln", "layernorm_bf16", "gemm_qkv",
    "attention", "gemm_out_proj_residual", "gemm_fc1_quickgelu", "gemm_fc2_residual", "pool_project",
    "last_block_cls_rows"};      // the last block's out_proj / fc1 / fc2 on the n CLS rows: a class of its own, so that the per-launch
                                 // averages of the three GEMM classes above are those of full-size launches only

// LayerNorm 1 / 2 are folded into the qkv / fc1 GEMMs (encoder_kernels.h "LayerNorm folded into the GEMMs"):
// w_qkv = g1 (.) W_qkv, c1_qkv[n] = sum_k w_qkv[n][k] (of the rounded 16-bit values), c2_qkv[n] = sum_k b1[k] W_qkv[n][k] + bias
struct LayerW {
    float *c1_qkv, *c2_qkv, *b_out, *c1_fc1, *c2_fc1, *b_fc2;
    uint16_t *w_qkv, *w_out, *w_fc1, *w_fc2;
};

struct Arena {            // one hipMalloc, 256-B aligned bump allocation
    char* base = nullptr; size_t size = 0, used = 0;
    template <class T> T* take(size_t count) {
        used = (used + 255) & ~(size_t)255;
        T* p = (T*)(base + used);
        used += count * sizeof(T);
        return p;
    }
};

}  // namespace vq

using namespace vq;

struct vq_encoder {
    vq_vit_config cfg{};
    int tokens = 0, patches = 0, grid = 0, patch_k = 0, max_batch = 0;
    int64_t rows_pad = 0, prow_pad = 0;
    hipStream_t stream = nullptr;       // stream in use
    hipStream_t own_stream = nullptr;   // created by the handle
    std::mutex mu;
    Arena arena;
    std::shared_ptr<void> arena_owner;     // frees arena.base when the last handle using it goes
    std::shared_ptr<void> weights_owner;   // vq_encoder_create_shared: the parent's arena, where this handle's weights live
    // weights
    uint16_t* w_patch = nullptr; float *b_patch = nullptr, *cls = nullptr, *pos = nullptr;
    float *pre_g = nullptr, *pre_b = nullptr, *post_g = nullptr, *post_b = nullptr, *w_proj = nullptr;
    std::vector<LayerW> layers;
    // workspace
    uint8_t* d_frames = nullptr;
    float *x = nullptr, *d_out = nullptr;
    uint16_t *h = nullptr, *qkv = nullptr, *att = nullptr, *mlp = nullptr;      // h = xh: the residual stream rounded to 16 bits
    uint16_t* xl = nullptr;                    // [r04] the low half of the split residual stream, one fp8 byte per element (EpiBiasResidualLnF32 modes)
    bool split_resid = true;                   // $VQ_AMD_RESID=f32: every residual epilogue reads and writes the fp32 x (rounds 1-3; the A/B switch)
    float2* ps = nullptr;                      // LayerNorm row partials [hidden/64][rows_pad]
    uint8_t* h_stage[2] = {nullptr, nullptr};   // pinned staging slots (lazy)
    float* h_out_stage = nullptr;               // pinned result buffer
    // pipelined ingest (vq_encoder_submit_staged / wait_staged): per-slot device frames + pinned results, a copy
    // stream for the uploads, events ordering upload -> forward -> download per slot
    uint8_t* d_slot_frames[2] = {nullptr, nullptr};
    float* h_slot_out[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_fwd[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    int slot_n[2] = {0, 0};                     // frames in flight per slot (0 = idle)
    int run_layers = -1;
    int last_n = 0;
    bool is_text = false;       // CLIP text tower (vq_text_encoder_*): token embedding, causal attention, EOS pooling
    float* tok_emb = nullptr; int* d_ids = nullptr; int* d_rowidx = nullptr; int vocab = 0, eos_id = 0;
    bool attn_simple = false;   // $VQ_AMD_ATTN=simple: per-wave streaming attention (reference implementation of the wg one)
    bool attn_q64 = false;      // $VQ_AMD_ATTN=q64: 64 query rows per wave (the round-2 form) instead of 32 (A/B switch)
    bool attn_t64 = false;      // $VQ_AMD_ATTN=t64: the run-time-T single-tile kernel where attention_tile_kernel<T> would run (A/B switch)
    bool prune_last = true;  // last block on CLS rows only (outputs unchanged)
    float patch_unscale = 1.0f;   // 2^-s: undoes the power-of-two scale on fp16 patch weights (EpiPatchEmbedF32)
    int f16_mask = 0;        // per-GEMM-group operand type, DT_* bits (set = fp16, clear = bf16): create flags / $VQ_AMD_DTYPE
    bool h_is_f16 = false;   // type of what `h` holds right now (debug_read)
    int gemm24_mask = 0;     // $VQ_AMD_GEMM24: which full-batch GEMMs run the hand-scheduled four-wave kernel (gemm_asm256.h): 1 qkv, 2 out_proj, 4 fc1, 8 fc2, 16 patch
                             // embedding.  Default none: 16 % fewer cycles per K-tile, and the chip answers with a 14 % lower clock - frames/s equal within 1 %
                             // with three batches in flight, +1.3 % for fc2 on a lone handle, -1 % on ViT-L/14 (DESIGN.md §4 "Round 3" (5))
    int gemm_force = 0;      // $VQ_AMD_GEMM: 0 auto, 1 = 128x128 kernel only, 2 = 256x256 wherever it tiles, 6 = auto without 160-row tiles
    // profiling
    bool profiling = false;
    struct Ev { int cls; hipEvent_t a, b; };
    std::vector<Ev> events;
    std::vector<hipEvent_t> pool;
};

namespace {

struct Prof {             // brackets one launch with events when profiling
    vq_encoder* e; int cls; hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(vq_encoder* e) {
        if (!e->pool.empty()) { hipEvent_t ev = e->pool.back(); e->pool.pop_back(); return ev; }
        hipEvent_t ev; (void)hipEventCreate(&ev); return ev;
    }
    Prof(vq_encoder* enc, int c) : e(enc), cls(c) {
        if (e->profiling) { a = get(e); b = get(e); (void)hipEventRecord(a, e->stream); }
    }
    ~Prof() {
        if (e->profiling) { (void)hipEventRecord(b, e->stream); e->events.push_back({cls, a, b}); }
    }
};

size_t arena_bytes(const vq_vit_config& c, int tokens, int patches, int patch_k, int max_batch,
                   int64_t rows_pad, int64_t prow_pad) {
    size_t h = c.hidden, m = c.mlp, n = 0;
    auto add = [&](size_t bytes) { n = ((n + 255) & ~(size_t)255) + bytes; };
    add((size_t)h * patch_k * 2); add(h * 4); add(h * 4); add((size_t)tokens * h * 4);
    add(h * 4); add(h * 4); add(h * 4); add(h * 4); add((size_t)c.proj_dim * h * 4);
    for (int l = 0; l < c.layers; ++l) {
        add(3 * h * 4); add(3 * h * 4); add(h * 4); add(m * 4); add(m * 4); add(h * 4);
        add(3 * h * h * 2); add(h * h * 2); add(m * h * 2); add(h * m * 2);
    }
    add((size_t)max_batch * c.image_size * c.image_size * 3);
    add((size_t)LN_MAX_GRANULES * rows_pad * 8);
    add((size_t)rows_pad * h * 4); add((size_t)max_batch * c.proj_dim * 4);
    add((size_t)rows_pad * h * 2); add((size_t)rows_pad * h * 2); add((size_t)rows_pad * 3 * h * 2); add((size_t)rows_pad * h * 2);
    size_t mlp_elems = std::max((size_t)rows_pad * m, (size_t)prow_pad * patch_k);
    add(mlp_elems * 2);
    (void)patches;
    return n + 4096;
}

int upload_f32(float* dst, const float* src, size_t n) {
    VQ_HIP(hipMemcpy(dst, src, n * 4, hipMemcpyHostToDevice));
    return 0;
}
int upload_h16(uint16_t* dst, const float* src, size_t n, bool f16, float scale = 1.0f) {
    std::vector<uint16_t> tmp(n);
    if (f16) for (size_t i = 0; i < n; ++i) tmp[i] = __builtin_bit_cast(uint16_t, (_Float16)(src[i] * scale));
    else     for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_bf16_rne(src[i] * scale);
    VQ_HIP(hipMemcpy(dst, tmp.data(), n * 2, hipMemcpyHostToDevice));
    return 0;
}

// Operand type per GEMM group (bit set = fp16, clear = bf16; fp32 accumulation either way, same MFMA rate).
// A group = every 16-bit tensor that meets in one MFMA: the weights and the activations that multiply them.
//   DT_PATCH  patch pixels (exact in both types) x W_patch
//   DT_QKV    LN1 output h x W_qkv
//   DT_ATTN   q | k | v, softmax probabilities, attention output x W_out
//   DT_FC1    LN2 output h x W_fc1
//   DT_FC2    quick-GELU output x W_fc2
enum : int { DT_PATCH = 1, DT_QKV = 2, DT_ATTN = 4, DT_FC1 = 8, DT_FC2 = 16, DT_ALL = 31 };

// W' = gamma (.) W (scaled), rounded to the operand type; c1[n] = sum_k W'16[n][k]; c2[n] = scale (sum_k beta[k] W[n][k] + bias[n])
int upload_folded(uint16_t* dst, float* c1_dst, float* c2_dst, const float* W, const float* gamma, const float* beta,
                  const float* bias, size_t N, size_t K, bool f16, float scale = 1.0f) {
    std::vector<uint16_t> w16(N * K);
    std::vector<float> c1(N), c2(N);
    for (size_t n = 0; n < N; ++n) {
        double s1 = 0.0, s2 = 0.0;
        for (size_t k = 0; k < K; ++k) {
            const float wf = W[n * K + k] * scale * gamma[k];
            const uint16_t r = f16 ? __builtin_bit_cast(uint16_t, (_Float16)wf) : f32_to_bf16_rne(wf);
            w16[n * K + k] = r;
            s1 += f16 ? (double)(float)__builtin_bit_cast(_Float16, r) : (double)bf16_to_f32(r);
            s2 += (double)beta[k] * (double)W[n * K + k];
        }
        c1[n] = (float)s1;
        c2[n] = (float)((s2 + (double)bias[n]) * (double)scale);
    }
    VQ_HIP(hipMemcpy(dst, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(c1_dst, c1.data(), N * 4, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(c2_dst, c2.data(), N * 4, hipMemcpyHostToDevice));
    return 0;
}

// One transformer block's tensors (HF order: ln1.{w,b}, q.{w,b}, k.{w,b}, v.{w,b}, out.{w,b}, ln2.{w,b}, fc1.{w,b}, fc2.{w,b})
// -> device, with layer_norm1 folded into the fused q|k|v weights (q additionally pre-scaled by d_h^-0.5) and
// layer_norm2 into fc1.
int upload_layer(vq_encoder* e, LayerW& L, Arena& A, const float* const* weights, int& wi, size_t H, size_t M, float qscale) {
    const float *g1 = weights[wi], *b1 = weights[wi + 1];
    wi += 2;
    L.w_qkv = A.take<uint16_t>(3 * H * H);
    L.c1_qkv = A.take<float>(3 * H);
    L.c2_qkv = A.take<float>(3 * H);
    for (int part = 0; part < 3; ++part) {        // q, k, v
        const float s = part == 0 ? qscale : 1.0f;
        VQ_TRY(upload_folded(L.w_qkv + part * H * H, L.c1_qkv + part * H, L.c2_qkv + part * H, weights[wi], g1, b1, weights[wi + 1],
                             H, H, e->f16_mask & DT_QKV, s));
        wi += 2;
    }
    L.w_out = A.take<uint16_t>(H * H);    VQ_TRY(upload_h16(L.w_out, weights[wi++], H * H, e->f16_mask & DT_ATTN));
    L.b_out = A.take<float>(H);           VQ_TRY(upload_f32(L.b_out, weights[wi++], H));
    const float *g2 = weights[wi], *b2 = weights[wi + 1];
    wi += 2;
    L.w_fc1 = A.take<uint16_t>(M * H);
    L.c1_fc1 = A.take<float>(M);
    L.c2_fc1 = A.take<float>(M);
    VQ_TRY(upload_folded(L.w_fc1, L.c1_fc1, L.c2_fc1, weights[wi], g2, b2, weights[wi + 1], M, H, e->f16_mask & DT_FC1, VQ_GELU_FOLD ? QUICK_GELU_C : 1.0f));
    wi += 2;
    L.w_fc2 = A.take<uint16_t>(H * M);    VQ_TRY(upload_h16(L.w_fc2, weights[wi++], H * M, e->f16_mask & DT_FC2, VQ_GELU_FOLD ? 1.0f / QUICK_GELU_C : 1.0f));
    L.b_fc2 = A.take<float>(H);           VQ_TRY(upload_f32(L.b_fc2, weights[wi++], H));
    return 0;
}

template <class Fn> static inline int by_f16(bool f16, Fn&& fn) {
    return f16 ? fn(std::true_type{}) : fn(std::false_type{});
}
#define VQ_F16(tag) (decltype(tag)::value)

template <int NV>
int run_forward(vq_encoder* e, const uint8_t* d_frames, int n, int swap_rb, float* d_out_f32, uint16_t* d_out_f16,
                const int* d_ids = nullptr) {
    const vq_vit_config& c = e->cfg;
    hipStream_t st = e->stream;
    const int H = c.hidden, T = e->tokens;
    const int rows = n * T;
    const bool fP = e->f16_mask & DT_PATCH, fQ = e->f16_mask & DT_QKV, fA = e->f16_mask & DT_ATTN,
               f1 = e->f16_mask & DT_FC1, f2 = e->f16_mask & DT_FC2;
    // GEMM row counts are padded (the buffers are): to 256 when that adds < 6 % work, so the phased
    // 256x256 kernel applies; small batches keep 128-row granularity
    auto pad_rows = [](int r) {
        const int r256 = (int)round_up(r, G2_BM), r128 = (int)round_up(r, GEMM_BM);
        return (r256 - r128) * 16 <= r128 ? r256 : r128;
    };
    const int rows_gemm = pad_rows(rows);
    // per-GEMM row count: 160-row tiles where they occupy more CUs than 256-row tiles (gemm_mfma160.h)
    // the hand-scheduled four-wave kernel (gemm_asm256.h) for the GEMMs $VQ_AMD_GEMM24 names, where 256-row tiles fill half the chip
    auto use24 = [&](int bit, int M, int N, int K) {
        return (e->gemm24_mask & bit) && (e->gemm_force == 0 || e->gemm_force == 6) && M % G2_BM == 0 && N % G2_BN == 0 && K % (2 * G2_BK) == 0 &&
               (int64_t)(M / G2_BM) * (N / G2_BN) >= 128;
    };
    auto gemm_rows = [&](int N, int K, int bit = 0) {
        if (use24(bit, rows_gemm, N, K)) return rows_gemm;
        const int r160 = (int)round_up(rows, G5_BM);
        return (e->gemm_force == 0 && gemm_use160() && r160 <= e->rows_pad && prefer_tn160(r160, N, K)) ? r160 : rows_gemm;
    };
    auto gf = [&](int bit, int M, int N, int K) { return use24(bit, M, N, K) ? 24 : e->gemm_force; };
    const int prows = n * e->patches;
    const int prows_gemm = pad_rows(prows);
    const LnPartials part{e->ps, e->rows_pad};
    const int granules = H / 64;
    const float inv_h = 1.0f / (float)H;

    const int nl = e->run_layers < 0 ? c.layers : std::min(e->run_layers, c.layers);
    if (e->is_text) {
        Prof p(e, C_EMBED_FINISH);
        by_f16(fQ, [&](auto F) {
            hipLaunchKernelGGL((embed_tokens_kernel<NV, VQ_F16(F)>), dim3(cdiv(rows, 4)), dim3(256), 0, st, d_ids, e->tok_emb,
                               e->pos, e->x, e->h, part, rows, T, e->vocab);
            return 0;
        });
        hipLaunchKernelGGL(eos_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, d_ids, e->d_rowidx, n, T, e->eos_id);
    } else {
    {   // E1/E2 + im2col: uint8 frames -> 16-bit patch rows (aliases the MLP buffer)
        Prof p(e, C_PATCHIFY);
        by_f16(fP, [&](auto F) {
            if (c.patch_size % 8 == 0 && e->patch_k == 3 * c.patch_size * c.patch_size) {
                const int64_t total = (int64_t)n * c.image_size * (c.image_size / 8);
                const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
                hipLaunchKernelGGL(patchify_u8_kernel<VQ_F16(F)>, dim3(blocks), dim3(256), 0, st, d_frames, e->mlp, n,
                                   c.image_size, c.patch_size, swap_rb);
            } else {
                const int64_t total = (int64_t)n * e->patches * 3 * c.patch_size;
                const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
                hipLaunchKernelGGL(patchify_generic_kernel<VQ_F16(F)>, dim3(blocks), dim3(256), 0, st, d_frames, e->mlp, n,
                                   c.image_size, c.patch_size, e->patch_k, swap_rb);
            }
            return 0;
        });
    }
    {   // E3: patch-embedding conv as a GEMM, epilogue scatters into token rows + position embedding
        Prof p(e, C_GEMM_PATCH);
        VQ_TRY(by_f16(fP, [&](auto F) {
            return launch_gemm_auto<VQ_F16(F)>(st, e->mlp, e->patch_k, e->w_patch, e->patch_k, prows_gemm, H, e->patch_k,
                                               EpiPatchEmbedF32{e->x, H, e->b_patch, e->pos, e->patches, T, prows, e->patch_unscale}, gf(16, prows_gemm, H, e->patch_k));
        }));
    }
    {   // CLS row, pre_layrnorm (in place); xh + row partials for the LN1 folded into layer 0's qkv GEMM
        Prof p(e, C_EMBED_FINISH);
        by_f16(fQ, [&](auto F) {
            hipLaunchKernelGGL((embed_finish_kernel<NV, VQ_F16(F)>), dim3(cdiv(rows, 4)), dim3(256), 0, st, e->x, e->h, e->cls,
                               e->pos, e->pre_g, e->pre_b, part, rows, T, c.ln_eps);
            return 0;
        });
    }
    }
    e->h_is_f16 = fQ;
    // [r04] the residual stream at 16 + 16 bits between the residual epilogues (encoder_kernels.h): fp16 operands in every group
    // that writes xh, and not in layer-limited debug runs (vq_encoder_debug_read reads the fp32 x).  The first residual epilogue
    // reads the embedding kernel's fp32 x; the last one in front of a reader of x (pooling head / the CLS-only last block) writes it.
    const bool cls_last = !e->is_text && e->prune_last && e->run_layers < 0;
    const bool split = e->split_resid && fQ && f1 && e->run_layers < 0;
    bool stream_split = false;                                 // what the residual stream is held as right now
    auto resid_mode = [&](bool wants_f32_after) {
        if (!split) return (int)RS_F32;
        const int m = (stream_split ? RS_IN_SPLIT : 0) | (wants_f32_after ? RS_OUT_F32 : RS_OUT_SPLIT);
        stream_split = !wants_f32_after;
        return m;
    };
    for (int l = 0; l < nl; ++l) {
        const LayerW& L = e->layers[l];
        {   // E5/E6: LN1 (folded) + fused q|k|v projection (q pre-scaled by d_h^-0.5 through its weights) on xh
            Prof p(e, C_GEMM_QKV);
            VQ_TRY(by_f16(fQ, [&](auto F) {
                return by_f16(fA, [&](auto FO) {
                    return launch_gemm_auto<VQ_F16(F)>(st, e->h, H, L.w_qkv, H, rows_gemm, 3 * H, H,
                                                       EpiLnH16<VQ_F16(FO), false>{e->qkv, 3 * H, L.c2_qkv, L.c1_qkv, part, granules, inv_h, c.ln_eps},
                                                       gf(1, rows_gemm, 3 * H, H));
                });
            }));
        }
        {
            Prof p(e, C_ATTENTION);
            by_f16(fA, [&](auto F) {
                constexpr bool F16 = VQ_F16(F);
                if (e->is_text) {
                    const int q_tiles = cdiv(T, 64), q_groups = cdiv(q_tiles, 4);
                    hipLaunchKernelGGL((attention_stream_wg_kernel<F16, true>), dim3(n * c.heads * q_groups), dim3(256), 0, st,
                                       e->qkv, e->att, T, H, c.heads, q_tiles, q_groups);
                } else if (T == 50 && c.heads % 4 == 0 && !e->attn_t64) {   // ViT-B/32 at 224^2: the compile-time-T form
                    hipLaunchKernelGGL((attention_tile_kernel<F16, 50>), dim3(n * (c.heads / 4)), dim3(256), 0, st, e->qkv, e->att, H, c.heads);
                } else if (T <= 64 && c.heads % 4 == 0) {
                    hipLaunchKernelGGL(attention_t64_kernel<F16>, dim3(n * (c.heads / 4)), dim3(256), 0, st, e->qkv, e->att, T, H,
                                       c.heads);
                } else {
                    const int q_tiles = cdiv(T, 64), q_groups = cdiv(q_tiles, 4);
                    if (e->attn_simple) {
                        const int units = n * c.heads * q_tiles;
                        hipLaunchKernelGGL((attention_stream_kernel<F16, false>), dim3(cdiv(units, 4)), dim3(256), 0, st, e->qkv,
                                           e->att, T, H, c.heads, q_tiles, units);
                    } else if (!e->attn_q64) {                  // 32 query rows per wave, three waves per SIMD (encoder_kernels.h)
                        const int q_tiles32 = cdiv(T, 32), q_groups32 = cdiv(q_tiles32, 4);
                        hipLaunchKernelGGL((attention_stream_wg_kernel<F16, false, 2>), dim3(n * c.heads * q_groups32), dim3(256), 0, st,
                                           e->qkv, e->att, T, H, c.heads, q_tiles32, q_groups32);
                    } else {
                        hipLaunchKernelGGL((attention_stream_wg_kernel<F16, false>), dim3(n * c.heads * q_groups), dim3(256), 0, st,
                                           e->qkv, e->att, T, H, c.heads, q_tiles, q_groups);
                    }
                }
                return 0;
            });
        }
        // Only the CLS token of the last block is consumed (E8): its out_proj / LN2 / MLP run on the
        // n CLS rows instead of n*T rows (292.8 MMAC of 4408.8 per frame).  K/V and the
        // attention itself still cover every token.  $VQ_AMD_FULL_LAST_LAYER=1 disables the pruning.
        const bool cls_only = !e->is_text && e->prune_last && e->run_layers < 0 && l == c.layers - 1;   // debug runs keep every row
        if (cls_only) {
            const int crows = pad_rows(n);
            {   // attention rows of the CLS tokens -> compact operand (the q|k|v buffer is free now); x, xh (compact) out
                Prof p(e, C_LAST_CLS);
                hipLaunchKernelGGL(gather_rows_h16_kernel, dim3(cdiv(n * (H / 8), 256)), dim3(256), 0, st, e->att, e->qkv, n, H, T);
                VQ_TRY(by_f16(fA, [&](auto F) {
                    return by_f16(f1, [&](auto FO) {
                        return launch_gemm_auto<VQ_F16(F)>(st, e->qkv, H, L.w_out, H, crows, H, H,
                                                           EpiBiasResidualClsLnF32<VQ_F16(FO)>{e->x, H, T, L.b_out, n, e->h, part}, e->gemm_force);
                    });
                }));
                e->h_is_f16 = f1;
            }
            {
                Prof p(e, C_LAST_CLS);
                VQ_TRY(by_f16(f1, [&](auto F) {
                    return by_f16(f2, [&](auto FO) {
                        return launch_gemm_auto<VQ_F16(F)>(st, e->h, H, L.w_fc1, H, crows, c.mlp, H,
                                                           EpiLnH16<VQ_F16(FO), true>{e->mlp, c.mlp, L.c2_fc1, L.c1_fc1, part, granules, inv_h, c.ln_eps},
                                                           e->gemm_force);
                    });
                }));
            }
            {   // 2 x 6 output tiles over K = mlp: split-K so that ~100 workgroups share the long K loop; partial planes
                // live in the (now free) q|k|v buffer, summed in slice order by the reduce kernel
                Prof p(e, C_LAST_CLS);
                const int splits = (c.mlp % (8 * GEMM_BK) == 0 && (size_t)8 * crows * H * 2 <= (size_t)e->rows_pad * 3 * H) ? 8 : 0;
                if (splits && crows % GEMM_BM == 0 && e->gemm_force != 2 && e->gemm_force != 8) {
                    float* part = (float*)e->qkv;
                    const int64_t plane = (int64_t)crows * H;
                    VQ_TRY(by_f16(f2, [&](auto F) {
                        return launch_gemm_tn_splitk<VQ_F16(F)>(st, e->mlp, c.mlp, L.w_fc2, c.mlp, crows, H, c.mlp, splits,
                                                                EpiSplitKPartialF32{part, H, plane});
                    }));
                    hipLaunchKernelGGL(splitk_reduce_residual_cls_kernel, dim3(cdiv((int64_t)n * (H / 4), 256)), dim3(256), 0, st, part, plane,
                                       splits, e->x, L.b_fc2, H, T, n);
                } else {
                    VQ_TRY(by_f16(f2, [&](auto F) {
                        return launch_gemm_auto<VQ_F16(F)>(st, e->mlp, c.mlp, L.w_fc2, c.mlp, crows, H, c.mlp,
                                                           EpiBiasResidualClsF32{e->x, H, T, L.b_fc2, n}, e->gemm_force);
                    }));
                }
            }
            continue;
        }
        {   // out_proj + residual; writes xh and the row partials for the LN2 folded into fc1
            Prof p(e, C_GEMM_OUT);
            const int mode = resid_mode(false);
            VQ_TRY(by_f16(fA, [&](auto F) {
                return by_f16(f1, [&](auto FO) {
                    auto go = [&](auto epi) { return launch_gemm_auto<VQ_F16(F)>(st, e->att, H, L.w_out, H, gemm_rows(H, H, 2), H, H, epi, gf(2, rows_gemm, H, H)); };
                    if constexpr (VQ_F16(FO)) {
                        if (mode == RS_OUT_SPLIT) return go(EpiBiasResidualLnF32<0, true, RS_OUT_SPLIT>{e->x, H, L.b_out, e->h, part, e->xl});
                        if (mode == (RS_IN_SPLIT | RS_OUT_SPLIT)) return go(EpiBiasResidualLnF32<0, true, RS_IN_SPLIT | RS_OUT_SPLIT>{e->x, H, L.b_out, e->h, part, e->xl});
                    }
                    return go(EpiBiasResidualLnF32<0, VQ_F16(FO)>{e->x, H, L.b_out, e->h, part});
                });
            }));
            e->h_is_f16 = f1;
        }
        {   // E7: LN2 (folded) + fc1 + quick_gelu on xh
            Prof p(e, C_GEMM_FC1);
            VQ_TRY(by_f16(f1, [&](auto F) {
                return by_f16(f2, [&](auto FO) {
                    return launch_gemm_auto<VQ_F16(F)>(st, e->h, H, L.w_fc1, H, rows_gemm, c.mlp, H,
                                                       EpiLnH16<VQ_F16(FO), true>{e->mlp, c.mlp, L.c2_fc1, L.c1_fc1, part, granules, inv_h, c.ln_eps},
                                                       gf(4, rows_gemm, c.mlp, H));
                });
            }));
        }
        {   // fc2 + residual; writes xh and the row partials for the next block's folded LN1
            Prof p(e, C_GEMM_FC2);
            // the fp32 x is wanted behind this epilogue when the pooling head comes next, or the CLS-only last block
            const int mode = resid_mode(l == nl - 1 || (cls_last && l == c.layers - 2));
            VQ_TRY(by_f16(f2, [&](auto F) {
                return by_f16(fQ, [&](auto FO) {
                    auto go = [&](auto epi) { return launch_gemm_auto<VQ_F16(F)>(st, e->mlp, c.mlp, L.w_fc2, c.mlp, gemm_rows(H, c.mlp, 8), H, c.mlp, epi, gf(8, rows_gemm, H, c.mlp)); };
                    if constexpr (VQ_F16(FO)) {
                        if (mode == (RS_IN_SPLIT | RS_OUT_SPLIT)) return go(EpiBiasResidualLnF32<1, true, RS_IN_SPLIT | RS_OUT_SPLIT>{e->x, H, L.b_fc2, e->h, part, e->xl});
                        if (mode == (RS_IN_SPLIT | RS_OUT_F32)) return go(EpiBiasResidualLnF32<1, true, RS_IN_SPLIT | RS_OUT_F32>{e->x, H, L.b_fc2, e->h, part, e->xl});
                    }
                    return go(EpiBiasResidualLnF32<1, VQ_F16(FO)>{e->x, H, L.b_fc2, e->h, part});
                });
            }));
            e->h_is_f16 = fQ;
        }
    }
    {   // E8-E10
        Prof p(e, C_POOL);
        hipLaunchKernelGGL((pool_project_kernel<NV>), dim3(cdiv(n, POOL_IMGS), cdiv(c.proj_dim, POOL_CHUNK)), dim3(256), 0, st,
                           e->x, e->post_g, e->post_b, e->w_proj, d_out_f32, n, T, c.proj_dim, c.ln_eps,
                           e->is_text ? e->d_rowidx : (const int*)nullptr);
        hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, d_out_f32, d_out_f16, n, c.proj_dim);
    }
    VQ_HIP(hipGetLastError());
    e->last_n = n;
    return 0;
}

int forward(vq_encoder* e, const uint8_t* d_frames, int n, int swap_rb, float* d_out_f32, uint16_t* d_out_f16,
            const int* d_ids = nullptr) {
    switch (e->cfg.hidden / 256) {
        case 2: return run_forward<2>(e, d_frames, n, swap_rb, d_out_f32, d_out_f16, d_ids);
        case 3: return run_forward<3>(e, d_frames, n, swap_rb, d_out_f32, d_out_f16, d_ids);
        case 4: return run_forward<4>(e, d_frames, n, swap_rb, d_out_f32, d_out_f16, d_ids);
        default: return fail(VQ_ERR_INVALID, "unsupported hidden size %d", e->cfg.hidden);
    }
}

// create flags / $VQ_AMD_DTYPE -> DT_* mask.  VQ_ENC_FP16: every group; VQ_ENC_F16_* bits: that group.
// $VQ_AMD_DTYPE = bf16 | fp16 | mixed | mask:<int> overrides the flags (experiments, bench.py --dtype).
int dtype_mask_from(int flags) {
    int mask = (flags & VQ_ENC_FP16) ? DT_ALL : ((flags >> 8) & DT_ALL);
    if (const char* dt = getenv("VQ_AMD_DTYPE")) {
        if (!strcmp(dt, "fp16") || !strcmp(dt, "f16")) mask = DT_ALL;
        else if (!strcmp(dt, "bf16")) mask = 0;
        else if (!strcmp(dt, "mixed")) mask = (VQ_ENC_MIXED >> 8) & DT_ALL;
        else if (!strncmp(dt, "mask:", 5)) mask = atoi(dt + 5) & DT_ALL;
    }
    return mask;
}

}  // namespace

extern "C" {

int vq_encoder_create(const vq_vit_config* cfg, const float* const* weights, int n_weights, int max_batch,
                      vq_encoder** out) {
    return vq_encoder_create_ex(cfg, weights, n_weights, max_batch, 0, out);
}

int vq_encoder_create_ex(const vq_vit_config* cfg, const float* const* weights, int n_weights, int max_batch,
                         int flags, vq_encoder** out) {
    VQ_TRY(require_init());
    VQ_CHECK(cfg && weights && out, "vq_encoder_create: null argument");
    const vq_vit_config c = *cfg;
    VQ_CHECK(c.layers > 0 && n_weights == 5 + 16 * c.layers + 3, "vq_encoder_create: expected %d weight tensors, got %d",
             5 + 16 * c.layers + 3, n_weights);
    VQ_CHECK(max_batch > 0 && max_batch <= 8192, "vq_encoder_create: max_batch %d out of range", max_batch);
    VQ_CHECK(c.image_size > 0 && c.patch_size > 0 && c.image_size % c.patch_size == 0,
             "vq_encoder_create: image %d is not a multiple of patch %d", c.image_size, c.patch_size);
    const int grid = c.image_size / c.patch_size, patches = grid * grid, tokens = patches + 1;
    VQ_CHECK(tokens <= 4096, "vq_encoder_create: %d tokens is beyond what this build sizes for", tokens);
    VQ_CHECK(c.hidden % c.heads == 0 && c.hidden / c.heads == 64, "vq_encoder_create: head_dim must be 64");
    VQ_CHECK((c.hidden == 768 || c.hidden == 1024) && c.mlp % 128 == 0, "vq_encoder_create: hidden %d / mlp %d unsupported",
             c.hidden, c.mlp);
    VQ_CHECK(c.proj_dim > 0 && c.proj_dim <= 4096, "vq_encoder_create: proj_dim %d out of range", c.proj_dim);
    const int patch_k_raw = 3 * c.patch_size * c.patch_size;
    const int patch_k = (int)round_up(patch_k_raw, 2 * G2_BK);     // zero-padded K (ViT-L/14: 588 -> 640)

    vq_encoder* e = new vq_encoder();
    if (flags & VQ_ENC_CONCURRENT) e->gemm_force = 6;        // auto, without the 160-row tiles
    if (const char* gf = getenv("VQ_AMD_GEMM")) e->gemm_force = atoi(gf);
    if (const char* rs = getenv("VQ_AMD_RESID")) e->split_resid = strcmp(rs, "f32") != 0;
    #ifdef VQ_DIAG
    if (const char* gm = getenv("VQ_AMD_GEMM24")) e->gemm24_mask = atoi(gm);
#endif
    if (const char* at = getenv("VQ_AMD_ATTN")) { e->attn_simple = !strcmp(at, "simple"); e->attn_q64 = !strcmp(at, "q64"); e->attn_t64 = !strcmp(at, "t64"); }
    if (const char* fl = getenv("VQ_AMD_FULL_LAST_LAYER")) e->prune_last = atoi(fl) == 0;
    e->f16_mask = dtype_mask_from(flags);
    e->cfg = c; e->tokens = tokens; e->patches = patches; e->grid = grid; e->patch_k = patch_k; e->max_batch = max_batch;
    e->rows_pad = round_up((int64_t)max_batch * tokens + (G5_BM - 1), 256);     // room for 256- and 160-row padding
    e->prow_pad = round_up((int64_t)max_batch * patches, 256);
    auto cleanup = [&](int rc) { vq_encoder_destroy(e); return rc; };

    const size_t bytes = arena_bytes(c, tokens, patches, patch_k, max_batch, e->rows_pad, e->prow_pad);
    hipError_t he = hipMalloc((void**)&e->arena.base, bytes);
    if (he != hipSuccess) { delete e; return fail(VQ_ERR_OOM, "vq_encoder_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(he)); }
    e->arena.size = bytes;
    e->arena_owner = std::shared_ptr<void>(e->arena.base, [](void* p) { (void)hipFree(p); });
    he = hipMemset(e->arena.base, 0, bytes);
    if (he != hipSuccess) return cleanup(fail(VQ_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(he)));
    he = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (he != hipSuccess) return cleanup(fail(VQ_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(he)));
    e->stream = e->own_stream;

    Arena& A = e->arena;
    const size_t H = c.hidden, M = c.mlp;
    int wi = 0;
    int rc = 0;
#define UP(expr) do { rc = (expr); if (rc) return cleanup(rc); } while (0)
    // embeddings.  Patch weights absorb ToTensor (/255) and Normalize ((x-mean)/std): the
    // patchify kernel stores (pixel-128), so W' = W/(255 std_c), bias = sum W (128/255-mean_c)/std_c.
    static const double mean[3] = {0.48145466, 0.4578275, 0.40821073};
    static const double stdv[3] = {0.26862954, 0.26130258, 0.27577711};
    e->cls = A.take<float>(H);                UP(upload_f32(e->cls, weights[wi++], H));
    {
        const float* wp = weights[wi++];
        const int pp = c.patch_size * c.patch_size;
        std::vector<uint16_t> w16(H * patch_k);
        std::vector<float> bias(H);
        // fp16 patch weights: W/(255 std) ~ 1e-4 for trained and seeded weights alike, i.e. inside fp16's subnormal range
        // (< 6.1e-5 loses mantissa bits).  Scale by the power of two that puts the largest |W'| just under 2^14; the GEMM
        // epilogue multiplies the fp32 accumulator by 2^-s (exact).  bf16 has fp32's exponent range: no scale.
        int pshift = 0;
        if (e->f16_mask & DT_PATCH) {
            double amax = 0.0;
            for (size_t n = 0; n < H; ++n)
                for (int ch = 0; ch < 3; ++ch)
                    for (int i = 0; i < pp; ++i) amax = std::max(amax, std::fabs((double)wp[(n * 3 + ch) * pp + i]) / (255.0 * stdv[ch]));
            if (amax > 0.0 && std::isfinite(amax)) pshift = std::min(24, std::max(0, (int)std::floor(std::log2(16000.0 / amax))));
        }
        e->patch_unscale = (float)std::ldexp(1.0, -pshift);
        const double pscale = std::ldexp(1.0, pshift);
        for (size_t n = 0; n < H; ++n) {
            double b = 0.0;
            for (int ch = 0; ch < 3; ++ch)
                for (int i = 0; i < pp; ++i) {
                    const double w = wp[(n * 3 + ch) * pp + i];
                    w16[n * patch_k + ch * pp + i] = (e->f16_mask & DT_PATCH) ? __builtin_bit_cast(uint16_t, (_Float16)(float)(w * pscale / (255.0 * stdv[ch])))
                                                             : f32_to_bf16_rne((float)(w / (255.0 * stdv[ch])));
                    b += w * (128.0 / 255.0 - mean[ch]) / stdv[ch];
                }
            bias[n] = (float)b;
        }
        e->w_patch = A.take<uint16_t>(H * patch_k);
        he = hipMemcpy(e->w_patch, w16.data(), w16.size() * 2, hipMemcpyHostToDevice);
        if (he != hipSuccess) return cleanup(fail(VQ_ERR_HIP, "weight upload failed: %s", hipGetErrorString(he)));
        e->b_patch = A.take<float>(H);        UP(upload_f32(e->b_patch, bias.data(), H));
    }
    e->pos = A.take<float>((size_t)tokens * H);  UP(upload_f32(e->pos, weights[wi++], (size_t)tokens * H));
    e->pre_g = A.take<float>(H);              UP(upload_f32(e->pre_g, weights[wi++], H));
    e->pre_b = A.take<float>(H);              UP(upload_f32(e->pre_b, weights[wi++], H));
    e->layers.resize(c.layers);
    const float qscale = 1.0f / std::sqrt((float)(c.hidden / c.heads));   // 0.125: exact in bf16
    for (int l = 0; l < c.layers; ++l) UP(upload_layer(e, e->layers[l], A, weights, wi, H, M, qscale));
    e->post_g = A.take<float>(H);             UP(upload_f32(e->post_g, weights[wi++], H));
    e->post_b = A.take<float>(H);             UP(upload_f32(e->post_b, weights[wi++], H));
    e->w_proj = A.take<float>((size_t)c.proj_dim * H);
    {   // stored transposed [hidden][proj_dim]: coalesced reads in pool_project_kernel
        const float* wp = weights[wi++];
        std::vector<float> wt((size_t)c.proj_dim * H);
        for (int o = 0; o < c.proj_dim; ++o)
            for (size_t k = 0; k < H; ++k) wt[k * c.proj_dim + o] = wp[(size_t)o * H + k];
        UP(upload_f32(e->w_proj, wt.data(), wt.size()));
    }
#undef UP
    // workspace
    e->d_frames = A.take<uint8_t>((size_t)max_batch * c.image_size * c.image_size * 3);
    e->ps = A.take<float2>((size_t)LN_MAX_GRANULES * e->rows_pad);
    e->x = A.take<float>((size_t)e->rows_pad * H);
    e->d_out = A.take<float>((size_t)max_batch * c.proj_dim);
    e->h = A.take<uint16_t>((size_t)e->rows_pad * H);
    e->xl = A.take<uint16_t>(VQ_RESID_XL8 ? ((size_t)e->rows_pad * H + 1) / 2 : (size_t)e->rows_pad * H);     // one byte per element (fp8 low half)
    e->qkv = A.take<uint16_t>((size_t)e->rows_pad * 3 * H);
    e->att = A.take<uint16_t>((size_t)e->rows_pad * H);
    e->mlp = A.take<uint16_t>(std::max((size_t)e->rows_pad * M, (size_t)e->prow_pad * patch_k));
    if (A.used > A.size) return cleanup(fail(VQ_ERR_STATE, "arena overflow (%zu > %zu)", A.used, A.size));
    *out = e;
    return 0;
}

// A second handle on the SAME weights: own stream, own workspace, own profiling state.  What a caller that keeps
// several batches in flight needs (bench.py --streams, the ingest loop's alternating handles) without copying the
// 176 MB of weights per handle.  The weights stay alive until the last handle that uses them is destroyed.
int vq_encoder_create_shared(vq_encoder* parent, int max_batch, int flags, vq_encoder** out) {
    VQ_TRY(require_init());
    VQ_CHECK(parent && out && !parent->is_text, "vq_encoder_create_shared: needs an image-encoder handle");
    VQ_CHECK(max_batch > 0 && max_batch <= 8192, "vq_encoder_create_shared: max_batch %d out of range", max_batch);
    const vq_vit_config c = parent->cfg;
    vq_encoder* e = new vq_encoder();
    e->cfg = c; e->tokens = parent->tokens; e->patches = parent->patches; e->grid = parent->grid; e->patch_k = parent->patch_k;
    e->max_batch = max_batch;
    e->f16_mask = parent->f16_mask;                          // the weights are already stored in these types
    e->patch_unscale = parent->patch_unscale;
    e->gemm_force = (flags & VQ_ENC_CONCURRENT) ? 6 : 0;
    if (const char* gf = getenv("VQ_AMD_GEMM")) e->gemm_force = atoi(gf);
    if (const char* rs = getenv("VQ_AMD_RESID")) e->split_resid = strcmp(rs, "f32") != 0;
    #ifdef VQ_DIAG
    if (const char* gm = getenv("VQ_AMD_GEMM24")) e->gemm24_mask = atoi(gm);
#endif
    e->attn_simple = parent->attn_simple; e->attn_q64 = parent->attn_q64; e->attn_t64 = parent->attn_t64; e->prune_last = parent->prune_last;
    e->rows_pad = round_up((int64_t)max_batch * e->tokens + (G5_BM - 1), 256);
    e->prow_pad = round_up((int64_t)max_batch * e->patches, 256);
    e->weights_owner = parent->weights_owner ? parent->weights_owner : parent->arena_owner;   // a clone of a clone still pins the original weights
    e->w_patch = parent->w_patch; e->b_patch = parent->b_patch; e->cls = parent->cls; e->pos = parent->pos;
    e->pre_g = parent->pre_g; e->pre_b = parent->pre_b; e->post_g = parent->post_g; e->post_b = parent->post_b;
    e->w_proj = parent->w_proj; e->layers = parent->layers;
    auto cleanup = [&](int rc) { vq_encoder_destroy(e); return rc; };
    const size_t H = c.hidden, M = c.mlp;
    size_t bytes = 0;
    auto add = [&](size_t b) { bytes = ((bytes + 255) & ~(size_t)255) + b; };
    add((size_t)max_batch * c.image_size * c.image_size * 3);
    add((size_t)LN_MAX_GRANULES * e->rows_pad * 8);
    add((size_t)e->rows_pad * H * 4); add((size_t)max_batch * c.proj_dim * 4);
    add((size_t)e->rows_pad * H * 2); add((size_t)e->rows_pad * H * 2); add((size_t)e->rows_pad * 3 * H * 2); add((size_t)e->rows_pad * H * 2);
    add(std::max((size_t)e->rows_pad * M, (size_t)e->prow_pad * e->patch_k) * 2);
    bytes += 4096;
    hipError_t he = hipMalloc((void**)&e->arena.base, bytes);
    if (he != hipSuccess) { delete e; return fail(VQ_ERR_OOM, "vq_encoder_create_shared: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(he)); }
    e->arena.size = bytes;
    e->arena_owner = std::shared_ptr<void>(e->arena.base, [](void* p) { (void)hipFree(p); });
    he = hipMemset(e->arena.base, 0, bytes);
    if (he != hipSuccess) return cleanup(fail(VQ_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(he)));
    he = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (he != hipSuccess) return cleanup(fail(VQ_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(he)));
    e->stream = e->own_stream;
    Arena& A = e->arena;
    e->d_frames = A.take<uint8_t>((size_t)max_batch * c.image_size * c.image_size * 3);
    e->ps = A.take<float2>((size_t)LN_MAX_GRANULES * e->rows_pad);
    e->x = A.take<float>((size_t)e->rows_pad * H);
    e->d_out = A.take<float>((size_t)max_batch * c.proj_dim);
    e->h = A.take<uint16_t>((size_t)e->rows_pad * H);
    e->xl = A.take<uint16_t>(VQ_RESID_XL8 ? ((size_t)e->rows_pad * H + 1) / 2 : (size_t)e->rows_pad * H);     // one byte per element (fp8 low half)
    e->qkv = A.take<uint16_t>((size_t)e->rows_pad * 3 * H);
    e->att = A.take<uint16_t>((size_t)e->rows_pad * H);
    e->mlp = A.take<uint16_t>(std::max((size_t)e->rows_pad * M, (size_t)e->prow_pad * e->patch_k));
    if (A.used > A.size) return cleanup(fail(VQ_ERR_STATE, "arena overflow (%zu > %zu)", A.used, A.size));
    *out = e;
    return 0;
}

int vq_text_encoder_create(const vq_text_config* cfg, const float* const* weights, int n_weights, int max_batch,
                           int flags, vq_text_encoder** out) {
    VQ_TRY(require_init());
    VQ_CHECK(cfg && weights && out, "vq_text_encoder_create: null argument");
    const vq_text_config t = *cfg;
    VQ_CHECK(t.layers > 0 && n_weights == 2 + 16 * t.layers + 3, "vq_text_encoder_create: expected %d weight tensors, got %d",
             2 + 16 * t.layers + 3, n_weights);
    VQ_CHECK(max_batch > 0 && max_batch <= 8192, "vq_text_encoder_create: max_batch %d out of range", max_batch);
    VQ_CHECK(t.hidden % t.heads == 0 && t.hidden / t.heads == 64, "vq_text_encoder_create: head_dim must be 64");
    VQ_CHECK((t.hidden == 512 || t.hidden == 768 || t.hidden == 1024) && t.mlp % 256 == 0,
             "vq_text_encoder_create: hidden %d / mlp %d unsupported", t.hidden, t.mlp);
    VQ_CHECK(t.max_positions > 0 && t.max_positions <= 4096 && t.vocab > 0, "vq_text_encoder_create: bad vocabulary/positions");
    VQ_CHECK(t.proj_dim > 0 && t.proj_dim <= 4096, "vq_text_encoder_create: proj_dim %d out of range", t.proj_dim);

    vq_encoder* e = new vq_encoder();
    e->is_text = true;
    e->cfg = vq_vit_config{0, 0, t.hidden, t.mlp, t.layers, t.heads, t.proj_dim, t.ln_eps};
    e->tokens = t.max_positions; e->patches = 0; e->grid = 0; e->patch_k = 0; e->max_batch = max_batch;
    e->vocab = t.vocab; e->eos_id = t.eos_token_id;
    e->rows_pad = round_up((int64_t)max_batch * e->tokens + (G5_BM - 1), 256);
    e->prow_pad = 0;
    if (const char* gf = getenv("VQ_AMD_GEMM")) e->gemm_force = atoi(gf);
    if (const char* rs = getenv("VQ_AMD_RESID")) e->split_resid = strcmp(rs, "f32") != 0;
    #ifdef VQ_DIAG
    if (const char* gm = getenv("VQ_AMD_GEMM24")) e->gemm24_mask = atoi(gm);
#endif
    e->f16_mask = dtype_mask_from(flags);
    auto cleanup = [&](int rc) { vq_encoder_destroy(e); return rc; };

    const size_t H = t.hidden, M = t.mlp, T = e->tokens;
    size_t bytes = 0;
    auto add = [&](size_t b) { bytes = ((bytes + 255) & ~(size_t)255) + b; };
    add((size_t)t.vocab * H * 4); add(T * H * 4); add(H * 4); add(H * 4); add((size_t)t.proj_dim * H * 4);
    for (int l = 0; l < t.layers; ++l) {
        add(3 * H * 4); add(3 * H * 4); add(H * 4); add(M * 4); add(M * 4); add(H * 4);
        add(3 * H * H * 2); add(H * H * 2); add(M * H * 2); add(H * M * 2);
    }
    add((size_t)LN_MAX_GRANULES * e->rows_pad * 8);
    add((size_t)max_batch * T * 4); add((size_t)max_batch * 4);
    add((size_t)e->rows_pad * H * 4); add((size_t)max_batch * t.proj_dim * 4);
    add((size_t)e->rows_pad * H * 2); add((size_t)e->rows_pad * H * 2); add((size_t)e->rows_pad * 3 * H * 2); add((size_t)e->rows_pad * H * 2);
    add((size_t)e->rows_pad * M * 2);
    bytes += 4096;
    hipError_t he = hipMalloc((void**)&e->arena.base, bytes);
    if (he != hipSuccess) { delete e; return fail(VQ_ERR_OOM, "vq_text_encoder_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(he)); }
    e->arena.size = bytes;
    e->arena_owner = std::shared_ptr<void>(e->arena.base, [](void* p) { (void)hipFree(p); });
    he = hipMemset(e->arena.base, 0, bytes);
    if (he != hipSuccess) return cleanup(fail(VQ_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(he)));
    he = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (he != hipSuccess) return cleanup(fail(VQ_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(he)));
    e->stream = e->own_stream;

    Arena& A = e->arena;
    int wi = 0, rc = 0;
#define UP(expr) do { rc = (expr); if (rc) return cleanup(rc); } while (0)
    e->tok_emb = A.take<float>((size_t)t.vocab * H);   UP(upload_f32(e->tok_emb, weights[wi++], (size_t)t.vocab * H));
    e->pos = A.take<float>(T * H);                      UP(upload_f32(e->pos, weights[wi++], T * H));
    e->layers.resize(t.layers);
    const float qscale = 1.0f / std::sqrt((float)(t.hidden / t.heads));
    for (int l = 0; l < t.layers; ++l) UP(upload_layer(e, e->layers[l], A, weights, wi, H, M, qscale));
    e->post_g = A.take<float>(H);             UP(upload_f32(e->post_g, weights[wi++], H));     // final_layer_norm
    e->post_b = A.take<float>(H);             UP(upload_f32(e->post_b, weights[wi++], H));
    e->w_proj = A.take<float>((size_t)t.proj_dim * H);
    {
        const float* wp = weights[wi++];
        std::vector<float> wt((size_t)t.proj_dim * H);
        for (int o = 0; o < t.proj_dim; ++o)
            for (size_t k = 0; k < H; ++k) wt[k * t.proj_dim + o] = wp[(size_t)o * H + k];
        UP(upload_f32(e->w_proj, wt.data(), wt.size()));
    }
#undef UP
    e->d_ids = A.take<int>((size_t)max_batch * T);
    e->d_rowidx = A.take<int>(max_batch);
    e->ps = A.take<float2>((size_t)LN_MAX_GRANULES * e->rows_pad);
    e->x = A.take<float>((size_t)e->rows_pad * H);
    e->d_out = A.take<float>((size_t)max_batch * t.proj_dim);
    e->h = A.take<uint16_t>((size_t)e->rows_pad * H);
    e->xl = A.take<uint16_t>(VQ_RESID_XL8 ? ((size_t)e->rows_pad * H + 1) / 2 : (size_t)e->rows_pad * H);     // one byte per element (fp8 low half)
    e->qkv = A.take<uint16_t>((size_t)e->rows_pad * 3 * H);
    e->att = A.take<uint16_t>((size_t)e->rows_pad * H);
    e->mlp = A.take<uint16_t>((size_t)e->rows_pad * M);
    if (A.used > A.size) return cleanup(fail(VQ_ERR_STATE, "arena overflow (%zu > %zu)", A.used, A.size));
    *out = e;
    return 0;
}

int vq_text_encoder_encode_ids(vq_text_encoder* e, const int32_t* ids, int n, int seq_len, float* out) {
    VQ_TRY(require_init());
    VQ_CHECK(e && e->is_text, "vq_text_encoder_encode_ids: not a text encoder handle");
    VQ_CHECK(n >= 0 && (n == 0 || (ids && out)), "vq_text_encoder_encode_ids: bad argument");
    VQ_CHECK(seq_len > 0 && seq_len <= e->tokens, "vq_text_encoder_encode_ids: seq_len %d outside (0, %d]", seq_len, e->tokens);
    std::lock_guard<std::mutex> lk(e->mu);
    const int T = e->tokens;
    std::vector<int32_t> padded;
    for (int done = 0; done < n; done += e->max_batch) {
        const int cur = std::min(e->max_batch, n - done);
        padded.assign((size_t)cur * T, e->eos_id);                       // pad with eos: invisible to the EOS position (causal)
        for (int i = 0; i < cur; ++i)
            std::copy(ids + (size_t)(done + i) * seq_len, ids + (size_t)(done + i + 1) * seq_len, padded.begin() + (size_t)i * T);
        VQ_HIP(hipMemcpyAsync(e->d_ids, padded.data(), padded.size() * 4, hipMemcpyHostToDevice, e->stream));
        VQ_TRY(forward(e, nullptr, cur, 0, e->d_out, nullptr, e->d_ids));
        VQ_HIP(hipMemcpyAsync(out + (size_t)done * e->cfg.proj_dim, e->d_out, (size_t)cur * e->cfg.proj_dim * 4,
                              hipMemcpyDeviceToHost, e->stream));
        VQ_HIP(hipStreamSynchronize(e->stream));
    }
    return 0;
}

int vq_text_encoder_destroy(vq_text_encoder* e) { return vq_encoder_destroy(e); }

int vq_encoder_destroy(vq_encoder* e) {
    if (!e) return 0;
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    for (auto& ev : e->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto ev : e->pool) (void)hipEventDestroy(ev);
    e->arena_owner.reset();          // frees the arena unless a shared handle still reads its weights
    e->weights_owner.reset();
    for (int i = 0; i < 2; ++i) if (e->h_stage[i]) (void)hipHostFree(e->h_stage[i]);
    if (e->h_out_stage) (void)hipHostFree(e->h_out_stage);
    if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
    for (int i = 0; i < 2; ++i) {
        if (e->d_slot_frames[i]) (void)hipFree(e->d_slot_frames[i]);
        if (e->h_slot_out[i]) (void)hipHostFree(e->h_slot_out[i]);
        if (e->ev_h2d[i]) (void)hipEventDestroy(e->ev_h2d[i]);
        if (e->ev_fwd[i]) (void)hipEventDestroy(e->ev_fwd[i]);
        if (e->ev_done[i]) (void)hipEventDestroy(e->ev_done[i]);
    }
    delete e;
    return 0;
}

int vq_encoder_output_dim(vq_encoder* e, int* dim) {
    VQ_CHECK(e && dim, "vq_encoder_output_dim: null argument");
    *dim = e->cfg.proj_dim;
    return 0;
}

int vq_encoder_encode_u8_device(vq_encoder* e, const void* d_frames, int n, int swap_rb, void* d_out_f32,
                                void* d_out_f16) {
    VQ_TRY(require_init());
    VQ_CHECK(e && d_frames && d_out_f32, "vq_encoder_encode_u8_device: null argument");
    VQ_CHECK(!e->is_text, "vq_encoder_encode_u8_device: this is a text encoder handle");
    VQ_CHECK(n > 0 && n <= e->max_batch, "vq_encoder_encode_u8_device: n=%d outside (0, max_batch=%d]", n, e->max_batch);
    std::lock_guard<std::mutex> lk(e->mu);
    return forward(e, (const uint8_t*)d_frames, n, swap_rb, (float*)d_out_f32, (uint16_t*)d_out_f16);
}

int vq_encoder_encode_u8(vq_encoder* e, const uint8_t* frames, int n, int swap_rb, float* out) {
    VQ_TRY(require_init());
    VQ_CHECK(e && n >= 0 && (n == 0 || (frames && out)), "vq_encoder_encode_u8: bad argument");
    VQ_CHECK(!e->is_text, "vq_encoder_encode_u8: this is a text encoder handle");
    std::lock_guard<std::mutex> lk(e->mu);
    const size_t fbytes = (size_t)e->cfg.image_size * e->cfg.image_size * 3;
    for (int done = 0; done < n; done += e->max_batch) {
        const int cur = std::min(e->max_batch, n - done);
        VQ_HIP(hipMemcpyAsync(e->d_frames, frames + (size_t)done * fbytes, cur * fbytes, hipMemcpyHostToDevice, e->stream));
        VQ_TRY(forward(e, e->d_frames, cur, swap_rb, e->d_out, nullptr));
        VQ_HIP(hipMemcpyAsync(out + (size_t)done * e->cfg.proj_dim, e->d_out, (size_t)cur * e->cfg.proj_dim * 4,
                              hipMemcpyDeviceToHost, e->stream));
        VQ_HIP(hipStreamSynchronize(e->stream));
    }
    return 0;
}

int vq_encoder_staging(vq_encoder* e, int slot, uint8_t** host_ptr, size_t* bytes) {
    VQ_TRY(require_init());
    VQ_CHECK(e && !e->is_text && host_ptr && bytes && (slot == 0 || slot == 1), "vq_encoder_staging: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    const size_t sz = (size_t)e->max_batch * e->cfg.image_size * e->cfg.image_size * 3;
    if (!e->h_stage[slot]) VQ_HIP(hipHostMalloc((void**)&e->h_stage[slot], sz));
    if (!e->h_out_stage) VQ_HIP(hipHostMalloc((void**)&e->h_out_stage, (size_t)e->max_batch * e->cfg.proj_dim * 4));
    *host_ptr = e->h_stage[slot];
    *bytes = sz;
    return 0;
}

int vq_encoder_encode_staged(vq_encoder* e, int slot, int n, int swap_rb, float* out) {
    VQ_TRY(require_init());
    VQ_CHECK(e && !e->is_text && out && (slot == 0 || slot == 1) && e->h_stage[slot], "vq_encoder_encode_staged: bad argument / slot not staged");
    VQ_CHECK(n > 0 && n <= e->max_batch, "vq_encoder_encode_staged: n=%d outside (0, max_batch=%d]", n, e->max_batch);
    std::lock_guard<std::mutex> lk(e->mu);
    const size_t fbytes = (size_t)e->cfg.image_size * e->cfg.image_size * 3;
    VQ_HIP(hipMemcpyAsync(e->d_frames, e->h_stage[slot], n * fbytes, hipMemcpyHostToDevice, e->stream));
    VQ_TRY(forward(e, e->d_frames, n, swap_rb, e->d_out, nullptr));
    VQ_HIP(hipMemcpyAsync(e->h_out_stage, e->d_out, (size_t)n * e->cfg.proj_dim * 4, hipMemcpyDeviceToHost, e->stream));
    VQ_HIP(hipStreamSynchronize(e->stream));
    memcpy(out, e->h_out_stage, (size_t)n * e->cfg.proj_dim * 4);
    return 0;
}

// Host gather: n separately allocated frames -> pinned staging slot, on up to n_threads threads.
int vq_encoder_stage_frames(vq_encoder* e, int slot, const uint8_t* const* frames, int n, int n_threads) {
    VQ_TRY(require_init());
    VQ_CHECK(e && !e->is_text && frames && (slot == 0 || slot == 1), "vq_encoder_stage_frames: bad argument");
    VQ_CHECK(n > 0 && n <= e->max_batch, "vq_encoder_stage_frames: n=%d outside (0, max_batch=%d]", n, e->max_batch);
    for (int i = 0; i < n; ++i) VQ_CHECK(frames[i], "vq_encoder_stage_frames: frame %d is null", i);
    uint8_t* dst = nullptr;
    size_t bytes = 0;
    VQ_TRY(vq_encoder_staging(e, slot, &dst, &bytes));
    {
        std::lock_guard<std::mutex> lk(e->mu);
        VQ_CHECK(e->slot_n[slot] == 0, "vq_encoder_stage_frames: slot %d still has a submitted batch (wait for it first)", slot);
    }
    const size_t fbytes = (size_t)e->cfg.image_size * e->cfg.image_size * 3;
    const int nt = std::max(1, std::min(std::min(n_threads, 16), n / 8));
    auto work = [&](int t) {
        for (int i = t; i < n; i += nt) memcpy(dst + (size_t)i * fbytes, frames[i], fbytes);
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    return 0;
}

// Asynchronous: upload slot -> forward -> download, ordered by events; the upload runs on the handle's copy
// stream so that it overlaps the forward pass of the other slot.
int vq_encoder_submit_staged(vq_encoder* e, int slot, int n, int swap_rb) {
    VQ_TRY(require_init());
    VQ_CHECK(e && !e->is_text && (slot == 0 || slot == 1) && e->h_stage[slot], "vq_encoder_submit_staged: bad argument / slot not staged");
    VQ_CHECK(n > 0 && n <= e->max_batch, "vq_encoder_submit_staged: n=%d outside (0, max_batch=%d]", n, e->max_batch);
    std::lock_guard<std::mutex> lk(e->mu);
    VQ_CHECK(e->slot_n[slot] == 0, "vq_encoder_submit_staged: slot %d already has a batch in flight", slot);
    const size_t fbytes = (size_t)e->cfg.image_size * e->cfg.image_size * 3;
    if (!e->copy_stream) VQ_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    if (!e->d_slot_frames[slot]) {
        VQ_HIP(hipMalloc((void**)&e->d_slot_frames[slot], (size_t)e->max_batch * fbytes));
        VQ_HIP(hipHostMalloc((void**)&e->h_slot_out[slot], (size_t)e->max_batch * e->cfg.proj_dim * 4));
        VQ_HIP(hipEventCreateWithFlags(&e->ev_h2d[slot], hipEventDisableTiming));
        VQ_HIP(hipEventCreateWithFlags(&e->ev_fwd[slot], hipEventDisableTiming));
        VQ_HIP(hipEventCreateWithFlags(&e->ev_done[slot], hipEventDisableTiming));
    } else {
        VQ_HIP(hipStreamWaitEvent(e->copy_stream, e->ev_fwd[slot], 0));    // the slot's previous forward has consumed its frames
    }
    VQ_HIP(hipMemcpyAsync(e->d_slot_frames[slot], e->h_stage[slot], n * fbytes, hipMemcpyHostToDevice, e->copy_stream));
    VQ_HIP(hipEventRecord(e->ev_h2d[slot], e->copy_stream));
    VQ_HIP(hipStreamWaitEvent(e->stream, e->ev_h2d[slot], 0));
    VQ_TRY(forward(e, e->d_slot_frames[slot], n, swap_rb, e->d_out, nullptr));
    VQ_HIP(hipEventRecord(e->ev_fwd[slot], e->stream));
    VQ_HIP(hipMemcpyAsync(e->h_slot_out[slot], e->d_out, (size_t)n * e->cfg.proj_dim * 4, hipMemcpyDeviceToHost, e->stream));
    VQ_HIP(hipEventRecord(e->ev_done[slot], e->stream));
    e->slot_n[slot] = n;
    return 0;
}

int vq_encoder_wait_staged(vq_encoder* e, int slot, float* out) {
    VQ_TRY(require_init());
    VQ_CHECK(e && out && (slot == 0 || slot == 1), "vq_encoder_wait_staged: bad argument");
    int n;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        n = e->slot_n[slot];
        VQ_CHECK(n > 0, "vq_encoder_wait_staged: slot %d has no batch in flight", slot);
    }
    VQ_HIP(hipEventSynchronize(e->ev_done[slot]));
    memcpy(out, e->h_slot_out[slot], (size_t)n * e->cfg.proj_dim * 4);
    std::lock_guard<std::mutex> lk(e->mu);
    e->slot_n[slot] = 0;
    return 0;
}

int vq_encoder_synchronize(vq_encoder* e) {
    VQ_CHECK(e, "vq_encoder_synchronize: null handle");
    VQ_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int vq_encoder_set_stream(vq_encoder* e, void* hip_stream) {
    VQ_CHECK(e, "vq_encoder_set_stream: null handle");
    std::lock_guard<std::mutex> lk(e->mu);
    VQ_HIP(hipStreamSynchronize(e->stream));
    e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    return 0;
}

int vq_encoder_profile_begin(vq_encoder* e) {
    VQ_CHECK(e, "vq_encoder_profile_begin: null handle");
    std::lock_guard<std::mutex> lk(e->mu);
    VQ_HIP(hipStreamSynchronize(e->stream));
    for (auto& ev : e->events) { e->pool.push_back(ev.a); e->pool.push_back(ev.b); }
    e->events.clear();
    e->profiling = true;
    return 0;
}

int vq_encoder_profile_end(vq_encoder* e, float* ms, int* launches) {
    VQ_CHECK(e && ms && launches, "vq_encoder_profile_end: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    e->profiling = false;
    VQ_HIP(hipStreamSynchronize(e->stream));
    for (int i = 0; i < VQ_ENC_NCLASS; ++i) { ms[i] = 0.f; launches[i] = 0; }
    for (auto& ev : e->events) {
        float t = 0.f;
        VQ_HIP(hipEventElapsedTime(&t, ev.a, ev.b));
        ms[ev.cls] += t; launches[ev.cls] += 1;
        e->pool.push_back(ev.a); e->pool.push_back(ev.b);
    }
    e->events.clear();
    return 0;
}

// What an event bracket measures beyond the kernel inside it: the median of 15 EMPTY brackets on the encoder's stream
// (start record, stop record, nothing between).  A bracket's elapsed time runs from the completion of the start marker to the
// completion of the stop marker, so it carries the marker-to-dispatch and completion-to-marker latencies of the command
// processor; rocprofv3's kernel durations (begin to end of the dispatch) do not.  bench.py reports both.
int vq_encoder_profile_bracket_overhead(vq_encoder* e, float* ms) {
    VQ_CHECK(e && ms, "vq_encoder_profile_bracket_overhead: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    constexpr int N = 15;
    hipEvent_t a[N], b[N];
    for (int i = 0; i < N; ++i) { a[i] = Prof::get(e); b[i] = Prof::get(e); }
    VQ_HIP(hipStreamSynchronize(e->stream));
    for (int i = 0; i < N; ++i) { VQ_HIP(hipEventRecord(a[i], e->stream)); VQ_HIP(hipEventRecord(b[i], e->stream)); }
    VQ_HIP(hipStreamSynchronize(e->stream));
    float t[N];
    for (int i = 0; i < N; ++i) { VQ_HIP(hipEventElapsedTime(&t[i], a[i], b[i])); e->pool.push_back(a[i]); e->pool.push_back(b[i]); }
    std::sort(t, t + N);
    *ms = t[N / 2];
    return 0;
}

const char* vq_encoder_profile_class_name(int cls) {
    return (cls >= 0 && cls < VQ_ENC_NCLASS) ? kEncClassNames[cls] : "";
}

int vq_encoder_debug_set_layers(vq_encoder* e, int layers) {
    VQ_CHECK(e, "vq_encoder_debug_set_layers: null handle");
    e->run_layers = layers;
    return 0;
}

int vq_encoder_debug_read(vq_encoder* e, const char* name, int rows, float* out) {
    VQ_CHECK(e && name && out, "vq_encoder_debug_read: null argument");
    VQ_CHECK(rows > 0 && rows <= e->rows_pad, "vq_encoder_debug_read: rows out of range");
    std::lock_guard<std::mutex> lk(e->mu);
    VQ_HIP(hipStreamSynchronize(e->stream));
    const size_t H = e->cfg.hidden;
    if (!strcmp(name, "x")) {
        VQ_HIP(hipMemcpy(out, e->x, (size_t)rows * H * 4, hipMemcpyDeviceToHost));
        return 0;
    }
    const uint16_t* src = nullptr; size_t cols = 0; bool f16 = false;
    if (!strcmp(name, "h")) { src = e->h; cols = H; f16 = e->h_is_f16; }
    else if (!strcmp(name, "qkv")) { src = e->qkv; cols = 3 * H; f16 = e->f16_mask & DT_ATTN; }
    else if (!strcmp(name, "att")) { src = e->att; cols = H; f16 = e->f16_mask & DT_ATTN; }
    else if (!strcmp(name, "mlp")) { src = e->mlp; cols = e->cfg.mlp; f16 = e->f16_mask & DT_FC2; }
    else return fail(VQ_ERR_INVALID, "vq_encoder_debug_read: unknown buffer '%s'", name);
    std::vector<uint16_t> tmp((size_t)rows * cols);
    VQ_HIP(hipMemcpy(tmp.data(), src, tmp.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i)
        out[i] = f16 ? (float)__builtin_bit_cast(_Float16, tmp[i]) : bf16_to_f32(tmp[i]);
    return 0;
}

}  // extern "C"

#ifdef VQ_GEMM_TOWER_STAMPS
// `make STAMPS=1` only: prints and clears the workgroup stamps gemm_tn256d_kernel collected (scripts/gemm_tower_stamps.py)
extern "C" int vq_debug_dump_gemm_stamps(void) {
    unsigned int n = 0;
    static unsigned long long h[4096 * 8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(vq::g_dbg_count), 4);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(vq::g_dbg_stamps), sizeof(h));
    if (n > 4096) n = 4096;
    for (unsigned i = 0; i < n; ++i) {
        const unsigned long long* d = h + i * 8;
        fprintf(stderr, "STAMP K %llu tn %llu epi %llu prologue %llu loop %llu epilogue %llu ticks %llu wg %llu\n", d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
    }
    const unsigned int z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vq::g_dbg_count), &z, 4);
    return (int)n;
}
#endif

// libvq_amd — frame preprocessing handle: Pillow-exact resize on the GPU in front of the
// encoder, and the frame-quality statistics of the reference's frame extractor.  See include/vq_amd.h.
#include "vq_common.h"
#include "preproc_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace vq {
int require_init();

// ---- Resample.c precompute_coeffs + normalize_coeffs_8bpc, in IEEE double like the C original ------------
static double filt_bilinear(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
static double filt_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

struct Coeffs {
    int ksize = 0;
    std::vector<int> bounds;     // [out][2]: first tap, tap count
    std::vector<int> kk;         // [out][ksize] fixed-point weights
};

static void precompute_coeffs(int in_size, int out_size, int filter, Coeffs& c) {
    double (*fn)(double) = filter == VQ_RESAMPLE_BICUBIC ? filt_bicubic : filt_bilinear;
    const double fsupport = filter == VQ_RESAMPLE_BICUBIC ? 2.0 : 1.0;
    const double in0 = 0.0, in1 = (double)in_size;
    double scale = (in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = fsupport * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;       // Resample.c's row length; rows here are padded to x4 with zeros
    c.ksize = (ksize + 3) & ~3;
    c.bounds.assign((size_t)out_size * 2, 0);
    c.kk.assign((size_t)out_size * c.ksize, 0);
    std::vector<double> w(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            w[x] = fn((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int* k = &c.kk[(size_t)xx * c.ksize];
        for (int x = 0; x < xmax; ++x) {
            double v = w[x];
            if (ww != 0.0) v /= ww;
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << RS_PRECISION_BITS)) : (int)(0.5 + v * (1 << RS_PRECISION_BITS));
        }
        c.bounds[2 * xx] = xmin;
        c.bounds[2 * xx + 1] = xmax;
    }
}

// OpenCV resize.cpp INTER_LINEAR tables for one axis: first tap and the two 11-bit weights per output index.
// snap_borders = the horizontal rule (weight 1 on the edge pixel outside [0, src-1)); rows are clipped by the kernel.
static void cv_linear_coeffs(int src, int dst, bool snap_borders, std::vector<int>& ofs, std::vector<int>& wts) {
    const double scale = (double)src / dst;
    ofs.resize(dst); wts.resize((size_t)dst * 2);
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (snap_borders) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= src - 1) { f = 0.f; s = src - 1; }
        }
        auto sat_short = [](float v) {                       // saturate_cast<short>(v): cvRound (half to even) + clamp
            long r = std::lrint(v);
            return (int)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
        };
        ofs[d] = s;
        wts[2 * d] = sat_short((1.f - f) * 2048.f);
        wts[2 * d + 1] = sat_short(f * 2048.f);
    }
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(VQ_ERR_OOM, "vq_resampler: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace vq

using namespace vq;

struct vq_resampler {
    std::mutex mu;
    hipStream_t stream = nullptr, own_stream = nullptr;
    DevBuf src, tmp, dst, coef, acc;
    // plan cache: coefficient tables on the device for the last geometry
    int p_h = 0, p_w = 0, p_filter = 0, p_out_h = 0, p_out_w = 0;
    Coeffs ch, cv;
    int *d_bh = nullptr, *d_kh = nullptr, *d_bv = nullptr, *d_kv = nullptr;
    int64_t out_bytes = 0;
};

namespace {

int plan(vq_resampler* r, int h, int w, int filter, int out_h, int out_w) {
    if (r->p_h == h && r->p_w == w && r->p_filter == filter && r->p_out_h == out_h && r->p_out_w == out_w) return 0;
    VQ_HIP(hipStreamSynchronize(r->stream));                 // earlier launches still read the old tables
    precompute_coeffs(w, out_w, filter, r->ch);
    precompute_coeffs(h, out_h, filter, r->cv);
    const size_t nbh = r->ch.bounds.size(), nkh = r->ch.kk.size(), nbv = r->cv.bounds.size(), nkv = r->cv.kk.size();
    VQ_TRY(r->coef.reserve((nbh + nkh + nbv + nkv) * sizeof(int)));
    r->d_bh = (int*)r->coef.p; r->d_kh = r->d_bh + nbh; r->d_bv = r->d_kh + nkh; r->d_kv = r->d_bv + nbv;
    VQ_HIP(hipMemcpyAsync(r->d_bh, r->ch.bounds.data(), nbh * 4, hipMemcpyHostToDevice, r->stream));
    VQ_HIP(hipMemcpyAsync(r->d_kh, r->ch.kk.data(), nkh * 4, hipMemcpyHostToDevice, r->stream));
    VQ_HIP(hipMemcpyAsync(r->d_bv, r->cv.bounds.data(), nbv * 4, hipMemcpyHostToDevice, r->stream));
    VQ_HIP(hipMemcpyAsync(r->d_kv, r->cv.kk.data(), nkv * 4, hipMemcpyHostToDevice, r->stream));
    VQ_HIP(hipStreamSynchronize(r->stream));                 // the host vectors may be rebuilt by the next plan
    r->p_h = h; r->p_w = w; r->p_filter = filter; r->p_out_h = out_h; r->p_out_w = out_w;
    return 0;
}

int check_geometry(const char* who, int n, int h, int w, int filter, int out_h, int out_w,
                   int crop_top, int crop_left, int crop_h, int crop_w) {
    VQ_CHECK(n >= 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0, "%s: sizes must be positive (n=%d h=%d w=%d out=%dx%d)",
             who, n, h, w, out_h, out_w);
    VQ_CHECK(filter == VQ_RESAMPLE_BILINEAR || filter == VQ_RESAMPLE_BICUBIC || filter == VQ_RESAMPLE_CV_LINEAR,
             "%s: filter %d is not BILINEAR(2), BICUBIC(3) or CV_LINEAR(100)", who, filter);
    VQ_CHECK(crop_h > 0 && crop_w > 0 && crop_top >= 0 && crop_left >= 0 && crop_top + crop_h <= out_h && crop_left + crop_w <= out_w,
             "%s: crop %dx%d at (%d,%d) does not fit the %dx%d resized frame", who, crop_h, crop_w, crop_top, crop_left, out_h, out_w);
    VQ_CHECK((int64_t)h * w < (int64_t)1 << 30, "%s: frame of %dx%d pixels is too large", who, h, w);
    VQ_CHECK(n <= 65535 && crop_h <= 65535 && h <= 65535 * RSH_ROWS, "%s: at most 65535 frames per call and 65535 output rows", who);
    return 0;
}

// both passes for n device-resident frames; d_dst gets [n][crop_h][crop_w][3]
// cv2.resize(frame, (out_w, out_h)) — INTER_LINEAR
int run_device_cv(vq_resampler* r, const uint8_t* d_src, int n, int h, int w, int out_h, int out_w,
                  int crop_top, int crop_left, int crop_h, int crop_w, uint8_t* d_dst) {
    const dim3 grid(cdiv(crop_w * 3, RS_THREADS), crop_h, n);
    if (out_h == h && out_w == w && crop_top == 0 && crop_left == 0 && crop_h == h && crop_w == w) {
        VQ_HIP(hipMemcpyAsync(d_dst, d_src, (size_t)n * h * w * 3, hipMemcpyDeviceToDevice, r->stream));
        return 0;
    }
    if (h == 2 * out_h && w == 2 * out_w) {
        hipLaunchKernelGGL(cv_resize_half_kernel, grid, dim3(RS_THREADS), 0, r->stream, d_src, d_dst, h, w, crop_top, crop_left,
                           crop_h, crop_w);
        VQ_HIP(hipGetLastError());
        return 0;
    }
    if (!(r->p_h == h && r->p_w == w && r->p_filter == VQ_RESAMPLE_CV_LINEAR && r->p_out_h == out_h && r->p_out_w == out_w)) {
        VQ_HIP(hipStreamSynchronize(r->stream));
        std::vector<int> xo, wx, yo, wy;
        cv_linear_coeffs(w, out_w, true, xo, wx);
        cv_linear_coeffs(h, out_h, false, yo, wy);
        const size_t total = xo.size() + wx.size() + yo.size() + wy.size();
        VQ_TRY(r->coef.reserve(total * sizeof(int)));
        r->d_bh = (int*)r->coef.p; r->d_kh = r->d_bh + xo.size(); r->d_bv = r->d_kh + wx.size(); r->d_kv = r->d_bv + yo.size();
        VQ_HIP(hipMemcpy(r->d_bh, xo.data(), xo.size() * 4, hipMemcpyHostToDevice));
        VQ_HIP(hipMemcpy(r->d_kh, wx.data(), wx.size() * 4, hipMemcpyHostToDevice));
        VQ_HIP(hipMemcpy(r->d_bv, yo.data(), yo.size() * 4, hipMemcpyHostToDevice));
        VQ_HIP(hipMemcpy(r->d_kv, wy.data(), wy.size() * 4, hipMemcpyHostToDevice));
        r->p_h = h; r->p_w = w; r->p_filter = VQ_RESAMPLE_CV_LINEAR; r->p_out_h = out_h; r->p_out_w = out_w;
    }
    hipLaunchKernelGGL(cv_resize_linear_kernel, grid, dim3(RS_THREADS), 0, r->stream, d_src, d_dst, r->d_bh, r->d_kh, r->d_bv,
                       r->d_kv, h, w, crop_top, crop_left, crop_h, crop_w);
    VQ_HIP(hipGetLastError());
    return 0;
}

int run_device(vq_resampler* r, const uint8_t* d_src, int n, int h, int w, int filter, int out_h, int out_w,
               int crop_top, int crop_left, int crop_h, int crop_w, uint8_t* d_dst) {
    if (filter == VQ_RESAMPLE_CV_LINEAR)
        return run_device_cv(r, d_src, n, h, w, out_h, out_w, crop_top, crop_left, crop_h, crop_w, d_dst);
    VQ_TRY(plan(r, h, w, filter, out_h, out_w));
    const int* bv = r->cv.bounds.data();
    const int row_first = bv[2 * crop_top];
    const int row_last = bv[2 * (crop_top + crop_h - 1)] + bv[2 * (crop_top + crop_h - 1) + 1];
    const int rows_needed = row_last - row_first;
    const size_t tmp_bytes = (size_t)n * rows_needed * crop_w * 3;
    if (tmp_bytes > r->tmp.cap) VQ_HIP(hipStreamSynchronize(r->stream));     // queued passes still use the old workspace
    VQ_TRY(r->tmp.reserve(tmp_bytes));
    // horizontal pass: 64 rows x `oc` output columns per workgroup; oc shrinks until the staged source span fits
    const int* bh = r->ch.bounds.data();
    auto span_max = [&](int oc) {                            // widest source span (pixels) of any segment
        int m = 0;
        for (int x0 = 0; x0 < crop_w; x0 += oc) {
            const int xa = crop_left + x0, xb = crop_left + std::min(x0 + oc, crop_w) - 1;
            m = std::max(m, bh[2 * xb] + bh[2 * xb + 1] - bh[2 * xa]);
        }
        return m;
    };
    int oc = 32, pitch_dw = 0, tile_pitch = 0;
    size_t lds = 0;
    // the row-block-major kernel (resample_hx_kernel) needs a staged row to be one wave-wide request of 16-byte loads
    // (<= 1 KiB with its alignment slack) and a segment's weights to fit four registers per thread
    static const bool no_hx = []() { const char* e = getenv("VQ_AMD_RSH"); return e && !strcmp(e, "seg"); }();      // A/B switch
    bool hx = !no_hx;
    for (int pass = 0; pass < 2; ++pass) {
        oc = 32;
        const int slack = hx ? 48 : 24;                      // hx: 15 bytes of alignment instead of 3
        for (;;) {
            pitch_dw = (span_max(oc) * 3 + slack + 3) / 4 | 1;    // + realignment and zero-weight over-read slack; odd
            tile_pitch = (oc * 3 + 3) / 4 * 4;
            if (((tile_pitch / 4) & 1) == 0) tile_pitch += 4;
            lds = (size_t)RSH_ROWS * pitch_dw * 4 + (size_t)RSH_ROWS * tile_pitch + (size_t)oc * r->ch.ksize * 4;
            // LDS per workgroup: 40 KB = four workgroups per CU for the row-block-major kernel (measured 32 / 40 / 48 / 64 KB:
            // 0.184 / 0.161 / 0.174 / 0.184 ms per 64 1080p frames), 48 KB for the segment-major one; $VQ_AMD_RSH_LDS_KB overrides
            static const int lds_kb = []() { const char* e = getenv("VQ_AMD_RSH_LDS_KB"); return e ? atoi(e) : 0; }();
            const size_t lds_cap = (size_t)(lds_kb > 0 ? lds_kb : hx ? 40 : 48) * 1024;
            const bool fits = lds <= lds_cap && (!hx || (pitch_dw * 4 <= 1024 && oc * r->ch.ksize <= 4 * RS_THREADS));
            if (fits || oc == 1) break;
            oc = oc > 4 ? oc - 4 : oc - 1;
        }
        if (!hx || (pitch_dw * 4 <= 1024 && oc * r->ch.ksize <= 4 * RS_THREADS && lds <= 96 * 1024)) break;
        hx = false;                                          // a single output column already spans more than a request: segment-major kernel
    }
    VQ_CHECK(lds <= 150 * 1024, "vq_resampler: a %d -> %d pixel row needs %zu bytes of LDS per workgroup", w, out_w, lds);
    const bool dword_store = (crop_w % 4 == 0) && (oc % 4 == 0);
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)resample_h_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        VQ_HIP(hipFuncSetAttribute((const void*)resample_h_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        VQ_HIP(hipFuncSetAttribute((const void*)resample_hx_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        VQ_HIP(hipFuncSetAttribute((const void*)resample_hx_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    if (hx) {
        // segments per workgroup: walk as many as leaves ~4+ rounds of workgroups on the chip
        const int nseg = cdiv(crop_w, oc), yblocks = cdiv(rows_needed, RSH_ROWS);
        static const int spw_env = []() { const char* e = getenv("VQ_AMD_RSH_SPW"); return e ? atoi(e) : 0; }();
        int xsplit = (int)std::max<int64_t>(1, std::min<int64_t>(nseg, cdiv((int64_t)4 * 768, (int64_t)yblocks * n)));
        int spw = spw_env > 0 ? std::min(spw_env, nseg) : cdiv(nseg, xsplit);
        const dim3 xgrid(cdiv(nseg, spw), yblocks, n);
        if (dword_store)
            hipLaunchKernelGGL(resample_hx_kernel<true>, xgrid, dim3(RS_THREADS), lds, r->stream, d_src, (uint8_t*)r->tmp.p,
                               r->d_bh, r->d_kh, r->ch.ksize, h, w, row_first, rows_needed, crop_left, crop_w, oc, pitch_dw, tile_pitch, spw);
        else
            hipLaunchKernelGGL(resample_hx_kernel<false>, xgrid, dim3(RS_THREADS), lds, r->stream, d_src, (uint8_t*)r->tmp.p,
                               r->d_bh, r->d_kh, r->ch.ksize, h, w, row_first, rows_needed, crop_left, crop_w, oc, pitch_dw, tile_pitch, spw);
    } else {
    const dim3 hgrid(cdiv(crop_w, oc), cdiv(rows_needed, RSH_ROWS), n);
    if (dword_store)
        hipLaunchKernelGGL(resample_h_kernel<true>, hgrid, dim3(RS_THREADS), lds, r->stream, d_src, (uint8_t*)r->tmp.p,
                           r->d_bh, r->d_kh, r->ch.ksize, h, w, row_first, rows_needed, crop_left, crop_w, oc, pitch_dw, tile_pitch);
    else
        hipLaunchKernelGGL(resample_h_kernel<false>, hgrid, dim3(RS_THREADS), lds, r->stream, d_src, (uint8_t*)r->tmp.p,
                           r->d_bh, r->d_kh, r->ch.ksize, h, w, row_first, rows_needed, crop_left, crop_w, oc, pitch_dw, tile_pitch);
    }
    VQ_HIP(hipGetLastError());
    const int row_bytes = crop_w * 3;
    if (row_bytes % 16 == 0 && ((uintptr_t)d_dst & 15) == 0 && ((uintptr_t)r->tmp.p & 15) == 0) {
        hipLaunchKernelGGL(resample_v_kernel<16>, dim3(cdiv((row_bytes / 16) * crop_h, RS_THREADS), 1, n), dim3(RS_THREADS), 0, r->stream,
                           (const uint8_t*)r->tmp.p, d_dst, r->d_bv, r->d_kv, r->cv.ksize, rows_needed, row_bytes, crop_top,
                           crop_h, row_first);
    } else if (row_bytes % 4 == 0 && ((uintptr_t)d_dst & 3) == 0) {
        hipLaunchKernelGGL(resample_v_kernel<4>, dim3(cdiv(row_bytes / 4, RS_THREADS), crop_h, n), dim3(RS_THREADS), 0, r->stream,
                           (const uint8_t*)r->tmp.p, d_dst, r->d_bv, r->d_kv, r->cv.ksize, rows_needed, row_bytes, crop_top,
                           crop_h, row_first);
    } else {
        hipLaunchKernelGGL(resample_v_kernel<1>, dim3(cdiv(row_bytes, RS_THREADS), crop_h, n), dim3(RS_THREADS), 0, r->stream,
                           (const uint8_t*)r->tmp.p, d_dst, r->d_bv, r->d_kv, r->cv.ksize, rows_needed, row_bytes, crop_top,
                           crop_h, row_first);
    }
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int vq_resampler_create(vq_resampler** out) {
    VQ_TRY(require_init());
    VQ_CHECK(out, "vq_resampler_create: null argument");
    vq_resampler* r = new vq_resampler();
    hipError_t e = hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete r; return fail(VQ_ERR_HIP, "vq_resampler_create: stream: %s", hipGetErrorString(e)); }
    r->stream = r->own_stream;
    *out = r;
    return 0;
}

int vq_resampler_destroy(vq_resampler* r) {
    if (!r) return 0;
    (void)hipStreamSynchronize(r->stream);
    r->src.release(); r->tmp.release(); r->dst.release(); r->coef.release(); r->acc.release();
    if (r->own_stream) (void)hipStreamDestroy(r->own_stream);
    delete r;
    return 0;
}

int vq_resampler_set_stream(vq_resampler* r, void* hip_stream) {
    VQ_CHECK(r, "vq_resampler_set_stream: null handle");
    std::lock_guard<std::mutex> lk(r->mu);
    VQ_HIP(hipStreamSynchronize(r->stream));
    r->stream = hip_stream ? (hipStream_t)hip_stream : r->own_stream;
    return 0;
}

int vq_resampler_synchronize(vq_resampler* r) {
    VQ_CHECK(r, "vq_resampler_synchronize: null handle");
    VQ_HIP(hipStreamSynchronize(r->stream));
    return 0;
}

int vq_resampler_run_u8_device(vq_resampler* r, const uint8_t* d_frames, int n, int h, int w, int filter,
                               int out_h, int out_w, int crop_top, int crop_left, int crop_h, int crop_w, uint8_t* d_out) {
    VQ_TRY(require_init());
    VQ_CHECK(r && (n == 0 || d_frames), "vq_resampler_run_u8_device: null argument");
    VQ_TRY(check_geometry("vq_resampler_run_u8_device", n, h, w, filter, out_h, out_w, crop_top, crop_left, crop_h, crop_w));
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(r->mu);
    const size_t ob = (size_t)n * crop_h * crop_w * 3;
    if (!d_out) {
        if (ob > r->dst.cap) VQ_HIP(hipStreamSynchronize(r->stream));     // a reader of the old buffer may be queued
        VQ_TRY(r->dst.reserve(ob));
        d_out = (uint8_t*)r->dst.p;
        r->out_bytes = (int64_t)ob;
    }
    return run_device(r, d_frames, n, h, w, filter, out_h, out_w, crop_top, crop_left, crop_h, crop_w, d_out);
}

int vq_resampler_run_u8_list(vq_resampler* r, const uint8_t* const* frames, int n, int h, int w, int filter,
                             int out_h, int out_w, int crop_top, int crop_left, int crop_h, int crop_w, uint8_t* out) {
    VQ_TRY(require_init());
    VQ_CHECK(r && (n == 0 || frames), "vq_resampler_run_u8_list: null argument");
    VQ_TRY(check_geometry("vq_resampler_run_u8_list", n, h, w, filter, out_h, out_w, crop_top, crop_left, crop_h, crop_w));
    if (n == 0) return 0;
    for (int i = 0; i < n; ++i) VQ_CHECK(frames[i], "vq_resampler_run_u8_list: frame %d is null", i);
    std::lock_guard<std::mutex> lk(r->mu);
    const size_t frame_bytes = (size_t)h * w * 3, ob_frame = (size_t)crop_h * crop_w * 3;
    VQ_HIP(hipStreamSynchronize(r->stream));
    VQ_TRY(r->dst.reserve((size_t)n * ob_frame));
    r->out_bytes = (int64_t)((size_t)n * ob_frame);
    // frames go up in slices of <= 512 MiB so the source workspace stays bounded
    int slice = (int)std::max<size_t>(1, ((size_t)512 << 20) / frame_bytes);
    if (slice > n) slice = n;
    VQ_TRY(r->src.reserve((size_t)slice * frame_bytes));
    for (int i = 0; i < n; i += slice) {
        const int m = std::min(slice, n - i);
        // consecutive frames that are contiguous in host memory go up in one copy
        for (int a = 0; a < m;) {
            int b = a + 1;
            while (b < m && frames[i + b] == frames[i + b - 1] + frame_bytes) ++b;
            VQ_HIP(hipMemcpyAsync((uint8_t*)r->src.p + (size_t)a * frame_bytes, frames[i + a], (size_t)(b - a) * frame_bytes,
                                  hipMemcpyHostToDevice, r->stream));
            a = b;
        }
        VQ_TRY(run_device(r, (const uint8_t*)r->src.p, m, h, w, filter, out_h, out_w, crop_top, crop_left, crop_h, crop_w,
                          (uint8_t*)r->dst.p + (size_t)i * ob_frame));
        VQ_HIP(hipStreamSynchronize(r->stream));             // the source slice is reused
    }
    if (out) {
        VQ_HIP(hipMemcpyAsync(out, r->dst.p, (size_t)n * ob_frame, hipMemcpyDeviceToHost, r->stream));
        VQ_HIP(hipStreamSynchronize(r->stream));
    }
    return 0;
}

int vq_resampler_run_u8(vq_resampler* r, const uint8_t* frames, int n, int h, int w, int filter,
                        int out_h, int out_w, int crop_top, int crop_left, int crop_h, int crop_w, uint8_t* out) {
    VQ_CHECK(r && n >= 0 && (n == 0 || frames) && h > 0 && w > 0, "vq_resampler_run_u8: bad argument");
    std::vector<const uint8_t*> ptrs((size_t)n);
    for (int i = 0; i < n; ++i) ptrs[i] = frames + (size_t)i * h * w * 3;
    return vq_resampler_run_u8_list(r, ptrs.data(), n, h, w, filter, out_h, out_w, crop_top, crop_left, crop_h, crop_w, out);
}

int vq_resampler_device_output(vq_resampler* r, void** d_ptr, int64_t* bytes) {
    VQ_CHECK(r && d_ptr, "vq_resampler_device_output: null argument");
    std::lock_guard<std::mutex> lk(r->mu);
    *d_ptr = r->dst.p;
    if (bytes) *bytes = r->out_bytes;
    return 0;
}

int vq_clip_processor_geometry(int h, int w, int size, int crop, int* resized_h, int* resized_w, int* crop_top, int* crop_left) {
    VQ_CHECK(h > 0 && w > 0 && size > 0 && crop > 0 && crop <= size && resized_h && resized_w && crop_top && crop_left,
             "vq_clip_processor_geometry: bad argument");
    // transformers image_transforms.py:295-299: the short edge becomes `size`, the long one int(size * long / short)
    const int shrt = w <= h ? w : h, lng = w <= h ? h : w;
    const int new_long = (int)((int64_t)size * lng / shrt);
    *resized_h = w <= h ? new_long : size;
    *resized_w = w <= h ? size : new_long;
    *crop_top = (*resized_h - crop) / 2;
    *crop_left = (*resized_w - crop) / 2;
    return 0;
}

int vq_frame_quality_u8(vq_resampler* r, const uint8_t* frames, int n, int h, int w, int on_device,
                        double* mean_brightness, double* laplacian_var) {
    VQ_TRY(require_init());
    VQ_CHECK(r && n >= 0 && h > 0 && w > 0 && (n == 0 || (frames && mean_brightness && laplacian_var)), "vq_frame_quality_u8: bad argument");
    VQ_CHECK((int64_t)h * w < (int64_t)1 << 30, "vq_frame_quality_u8: frame of %dx%d pixels is too large", h, w);
    VQ_CHECK(n <= 65535, "vq_frame_quality_u8: at most 65535 frames per call");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(r->mu);
    const size_t frame_bytes = (size_t)h * w * 3;
    VQ_HIP(hipStreamSynchronize(r->stream));
    VQ_TRY(r->acc.reserve((size_t)n * 3 * sizeof(long long)));
    VQ_HIP(hipMemsetAsync(r->acc.p, 0, (size_t)n * 3 * sizeof(long long), r->stream));
    int slice = n;
    if (!on_device) {
        slice = (int)std::max<size_t>(1, ((size_t)512 << 20) / frame_bytes);
        if (slice > n) slice = n;
        VQ_TRY(r->src.reserve((size_t)slice * frame_bytes));
    }
    const int wgs = std::min(cdiv((int64_t)h * w, RS_THREADS), 1024);
    for (int i = 0; i < n; i += slice) {
        const int m = std::min(slice, n - i);
        const uint8_t* d = frames + (size_t)i * frame_bytes;
        if (!on_device) {
            VQ_HIP(hipMemcpyAsync(r->src.p, d, (size_t)m * frame_bytes, hipMemcpyHostToDevice, r->stream));
            d = (const uint8_t*)r->src.p;
        }
        hipLaunchKernelGGL(frame_quality_kernel, dim3(wgs, m), dim3(RS_THREADS), 0, r->stream, d,
                           (long long*)r->acc.p + (size_t)i * 3, h, w);
        VQ_HIP(hipGetLastError());
        if (!on_device) VQ_HIP(hipStreamSynchronize(r->stream));
    }
    std::vector<long long> acc((size_t)n * 3);
    VQ_HIP(hipMemcpyAsync(acc.data(), r->acc.p, acc.size() * sizeof(long long), hipMemcpyDeviceToHost, r->stream));
    VQ_HIP(hipStreamSynchronize(r->stream));
    const double npix = (double)h * w;
    for (int i = 0; i < n; ++i) {
        mean_brightness[i] = (double)acc[3 * i] / (npix * 3.0);
        // var = E[L^2] - E[L]^2 from exact integer sums: (N*S2 - S1^2) / N^2
        const __int128 num = (__int128)((int64_t)h * w) * acc[3 * i + 2] - (__int128)acc[3 * i + 1] * acc[3 * i + 1];
        laplacian_var[i] = (double)((long double)num / ((long double)npix * (long double)npix));
    }
    return 0;
}

}  // extern "C"

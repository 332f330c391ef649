// Library-level entry points: device binding, error string, version, and the
// GEMM unit-test hook.
#include "../../include/vq_amd.h"
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "gemm_mfma256p.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace vq {

std::string& last_error() {
    static thread_local std::string msg;
    return msg;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

static int g_device = -1;

int require_init() {
    if (g_device < 0) return fail(VQ_ERR_STATE, "vq_init() has not been called (no device bound)");
    VQ_HIP(hipSetDevice(g_device));
    return 0;
}

struct EpiStoreF32 {
    float* out; int ldo;
    static constexpr bool kLoads = false;
    __device__ __forceinline__ f32x4 bias_at(int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ f32x4 load(int, int) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void store(int m, int n, f32x4 v, f32x4, f32x4) const {
        *(f32x4*)(out + (size_t)m * ldo + n) = v;
    }
};

}  // namespace vq

using namespace vq;

extern "C" {

const char* vq_last_error(void) { return last_error().c_str(); }
const char* vq_version(void) { return "vq_amd 0.1.0 (gfx950)"; }

int vq_device_count(int* count) {
    VQ_CHECK(count != nullptr, "vq_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return 0;
}

int vq_init(int device_ordinal) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(VQ_ERR_HIP, "vq_init: no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    VQ_CHECK(device_ordinal >= 0 && device_ordinal < n, "vq_init: device %d out of range [0,%d)", device_ordinal, n);
    // one process = one GPU (one rank per GPU): kernel attributes and handles are set up for the bound device only
    VQ_CHECK(g_device < 0 || g_device == device_ordinal, "vq_init: this process is already bound to device %d; "
             "use one process per GPU (torch.distributed.run / bench.py --gpus N)", g_device);
    VQ_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    VQ_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VQ_ERR_HIP, "vq_init: device %d is %s; the kernels are built for gfx950 only",
                    device_ordinal, prop.gcnArchName);
    g_device = device_ordinal;
    return 0;
}

int vq_debug_gemm(const float* A, const float* W, int M, int N, int K, int flags, float* C) {
    VQ_TRY(require_init());
    VQ_CHECK(A && W && C, "vq_debug_gemm: null argument");
    const int use_f16 = flags & 1, force = (flags >> 1) & 31;    // force: launch_gemm_auto's kernel ids (0 auto, 1 = 128x128, 2 = four-phase, 5 = 160x256 ring, 8 / 11 deep prefetch, 16 multi-tile, ...)
    std::vector<uint16_t> a16((size_t)M * K), w16((size_t)N * K);
    for (size_t i = 0; i < a16.size(); ++i)
        a16[i] = use_f16 ? __builtin_bit_cast(uint16_t, (_Float16)A[i]) : f32_to_bf16_rne(A[i]);
    for (size_t i = 0; i < w16.size(); ++i)
        w16[i] = use_f16 ? __builtin_bit_cast(uint16_t, (_Float16)W[i]) : f32_to_bf16_rne(W[i]);
    uint16_t *dA = nullptr, *dW = nullptr;
    float* dC = nullptr;
    VQ_HIP(hipMalloc(&dA, a16.size() * 2));
    VQ_HIP(hipMalloc(&dW, w16.size() * 2));
    VQ_HIP(hipMalloc(&dC, (size_t)M * N * 4));
    VQ_HIP(hipMemcpy(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(dW, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    int rc = use_f16 ? launch_gemm_auto<true>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N}, force)
                     : launch_gemm_auto<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N}, force);
    if (rc == 0) {
        hipError_t e = hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(VQ_ERR_HIP, "vq_debug_gemm: copy back failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC);
    return rc;
}

// Diagnostic: the 256x256 kernel with in-kernel s_memtime stamps (workgroup 0, 8 waves x 768 stamps).
int vq_debug_gemm_stamps(int M, int N, int K, int diag, unsigned long long* stamps /*[8][768]*/) {
    VQ_TRY(require_init());
    VQ_CHECK(stamps, "vq_debug_gemm_stamps: null argument");
    uint16_t *dA = nullptr, *dW = nullptr; float* dC = nullptr; unsigned long long* dS = nullptr;
    VQ_HIP(hipMalloc(&dA, (size_t)M * K * 2)); VQ_HIP(hipMalloc(&dW, (size_t)N * K * 2));
    VQ_HIP(hipMalloc(&dC, (size_t)M * N * 4)); VQ_HIP(hipMalloc(&dS, 8 * G2_MAX_STAMPS * 8));
    std::vector<uint16_t> a16((size_t)M * K), w16((size_t)N * K);
    uint32_t r = 12345;
    for (auto& v : a16) { r = r * 1664525u + 1013904223u; v = f32_to_bf16_rne(((int)(r >> 8) % 2001 - 1000) * 1e-3f); }
    for (auto& v : w16) { r = r * 1664525u + 1013904223u; v = f32_to_bf16_rne(((int)(r >> 8) % 2001 - 1000) * 1e-3f); }
    VQ_HIP(hipMemcpy(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(dW, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    VQ_HIP(hipMemset(dS, 0, 8 * G2_MAX_STAMPS * 8));
    int rc = 0;
    for (int rep = 0; rep < 3 && rc == 0; ++rep)
        rc = launch_gemm_tn256_stamped<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N}, dS, diag);
    if (rc == 0) {
        hipError_t e = hipMemcpy(stamps, dS, 8 * G2_MAX_STAMPS * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(VQ_ERR_HIP, "stamps copy failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); (void)hipFree(dS);
    return rc;
}

// Diagnostic: time one mainloop with parts removed (results invalid).  kernel 2 = 4-phase, 3 = ring.
// diag: bit0 no in-loop DMA, bit1 no ds_reads, bit2 no MFMAs, bit3 (ring only) no barriers.
int vq_debug_gemm_ablate(int M, int N, int K, int kernel, int diag, int reps, float* ms_avg) {
    VQ_TRY(require_init());
    VQ_CHECK(ms_avg && reps > 0, "vq_debug_gemm_ablate: bad argument");
    uint16_t *dA = nullptr, *dW = nullptr; float* dC = nullptr;
    VQ_HIP(hipMalloc(&dA, (size_t)M * K * 2)); VQ_HIP(hipMalloc(&dW, (size_t)N * K * 2)); VQ_HIP(hipMalloc(&dC, (size_t)M * N * 4));
    std::vector<uint16_t> a16((size_t)M * K), w16((size_t)N * K);
    uint32_t r = 777;
    for (auto& v : a16) { r = r * 1664525u + 1013904223u; v = f32_to_bf16_rne(((int)(r >> 8) % 2001 - 1000) * 1e-3f); }
    for (auto& v : w16) { r = r * 1664525u + 1013904223u; v = f32_to_bf16_rne(((int)(r >> 8) % 2001 - 1000) * 1e-3f); }
    VQ_HIP(hipMemcpy(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(dW, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    VQ_HIP(hipEventCreate(&e0)); VQ_HIP(hipEventCreate(&e1));
    int rc = 0;
    auto once = [&]() {
#ifdef VQ_GEMM_EXPERIMENTS
        if (kernel == 3) return launch_gemm_tn256_ring_diag<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N}, diag);
        if (kernel == 9) return launch_gemm_tn256e<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N});
        if (kernel == 10) return launch_gemm_tn256f<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N});
        if (kernel == 7) return launch_gemm_tn256w4<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N});
        if (kernel == 4) return launch_gemm_tn256p<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N});
#else
        if (kernel == 3 || kernel == 4 || kernel == 7 || kernel == 9 || kernel == 10)
            return fail(VQ_ERR_INVALID, "gemm kernel %d is an experiment: rebuild with `make EXPERIMENTS=1`", kernel);
#endif
        if (kernel == 8) return launch_gemm_tn256d<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N});
        if (kernel == 11) return launch_gemm_tn256d<false, EpiStoreF32, false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N});

        if (kernel == 1) return launch_gemm_tn<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N});
        if (kernel == 5) return launch_gemm_tn160_ring<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N});
        return launch_gemm_tn256_stamped<false>(nullptr, dA, K, dW, K, M, N, K, EpiStoreF32{dC, N}, nullptr, diag);
    };
    for (int i = 0; i < 3 && rc == 0; ++i) rc = once();
    VQ_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps && rc == 0; ++i) rc = once();
    VQ_HIP(hipEventRecord(e1, nullptr));
    VQ_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    VQ_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC);
    return rc;
}

// Diagnostic: the clock the chip holds inside the deep-prefetch mainloop (MI355X_MICROARCH.md "DVFS give-back" item 6):
// d s_memtime / d s_memrealtime x 100 MHz around the K loop, median over workgroups, after `reps` back-to-back launches.
int vq_debug_gemm_clock(int M, int N, int K, int reps, float* ms_avg, float* ghz_median) {
    VQ_TRY(require_init());
    VQ_CHECK(ms_avg && ghz_median && reps > 0 && M % 256 == 0 && N % 256 == 0 && K % 128 == 0, "vq_debug_gemm_clock: bad argument");
    uint16_t *dA = nullptr, *dW = nullptr; float* dC = nullptr; unsigned long long* dS = nullptr;
    const int wgs = (M / 256) * (N / 256);
    VQ_HIP(hipMalloc(&dA, (size_t)M * K * 2)); VQ_HIP(hipMalloc(&dW, (size_t)N * K * 2)); VQ_HIP(hipMalloc(&dC, (size_t)M * N * 4));
    VQ_HIP(hipMalloc(&dS, (size_t)wgs * 16));
    std::vector<uint16_t> a16((size_t)M * K), w16((size_t)N * K);
    uint32_t r = 4242;
    for (auto& v : a16) { r = r * 1664525u + 1013904223u; v = f32_to_bf16_rne(((int)(r >> 8) % 2001 - 1000) * 1e-3f); }
    for (auto& v : w16) { r = r * 1664525u + 1013904223u; v = f32_to_bf16_rne(((int)(r >> 8) % 2001 - 1000) * 1e-3f); }
    VQ_HIP(hipMemcpy(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(dW, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    typedef EpiStoreF32 E;
    VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256d_kernel<false, E, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
    hipEvent_t e0, e1;
    VQ_HIP(hipEventCreate(&e0)); VQ_HIP(hipEventCreate(&e1));
    auto once = [&]() {
        hipLaunchKernelGGL((gemm_tn256d_kernel<false, E, 1, true>), dim3(wgs), dim3(G2_THREADS), G2_LDS_BYTES, nullptr,
                           dA, K, dW, K, K, N / 256, E{dC, N}, 0, dS);
    };
    for (int i = 0; i < 3; ++i) once();
    VQ_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps; ++i) once();
    VQ_HIP(hipEventRecord(e1, nullptr));
    VQ_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    VQ_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / reps;
    std::vector<unsigned long long> st((size_t)wgs * 2);
    VQ_HIP(hipMemcpy(st.data(), dS, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<float> ghz;
    for (int i = 0; i < wgs; ++i) if (st[2 * i + 1]) ghz.push_back((float)st[2 * i] / (float)st[2 * i + 1] * 0.1f);
    std::sort(ghz.begin(), ghz.end());
    *ghz_median = ghz.empty() ? 0.f : ghz[ghz.size() / 2];
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); (void)hipFree(dS);
    return 0;
}

// Diagnostic: per-phase s_memtime stamps of workgroup 0 of the deep-prefetch mainloop (four per phase: phase start, before
// the mid barrier, before the MFMAs, after the MFMAs), random operands, after `reps` back-to-back launches.
int vq_debug_gemm_stamps_deep(int M, int N, int K, int reps, unsigned long long* stamps /*[8][512]*/) {
    VQ_TRY(require_init());
    VQ_CHECK(stamps && reps > 0 && M % 256 == 0 && N % 256 == 0 && K % 128 == 0, "vq_debug_gemm_stamps_deep: bad argument");
    uint16_t *dA = nullptr, *dW = nullptr; float* dC = nullptr; unsigned long long* dS = nullptr;
    VQ_HIP(hipMalloc(&dA, (size_t)M * K * 2)); VQ_HIP(hipMalloc(&dW, (size_t)N * K * 2)); VQ_HIP(hipMalloc(&dC, (size_t)M * N * 4));
    VQ_HIP(hipMalloc(&dS, (size_t)8 * G2D_STAMPS * 8));
    std::vector<uint16_t> a16((size_t)M * K), w16((size_t)N * K);
    uint32_t r = 999;
    for (auto& v : a16) { r = r * 1664525u + 1013904223u; v = f32_to_bf16_rne(((int)(r >> 8) % 2001 - 1000) * 1e-3f); }
    for (auto& v : w16) { r = r * 1664525u + 1013904223u; v = f32_to_bf16_rne(((int)(r >> 8) % 2001 - 1000) * 1e-3f); }
    VQ_HIP(hipMemcpy(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(dW, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    typedef EpiStoreF32 E;
    const int lds = G2_LDS_BYTES + 8 * G2D_STAMPS * 8;
    VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256d_kernel<false, E, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((gemm_tn256d_kernel<false, E, 2, true>), dim3((M / 256) * (N / 256)), dim3(G2_THREADS), lds, nullptr,
                           dA, K, dW, K, K, N / 256, E{dC, N}, 0, dS);
    VQ_HIP(hipDeviceSynchronize());
    VQ_HIP(hipMemcpy(stamps, dS, (size_t)8 * G2D_STAMPS * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); (void)hipFree(dS);
    return 0;
}

}  // extern "C"

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

}  // extern "C"

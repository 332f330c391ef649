// Diagnostic entry points (include/vq_amd_diag.h): in-kernel stamps, mainloop ablations, the clock inside the K loop, and
// vq_debug_gemm_bench — one GEMM kernel with one of the tower's epilogues timed in isolation.  NOT part of the product
// library: compiled only by `make DIAG=1` (implied by EXPERIMENTS=1 / STAMPS=1) into a library of its own that
// scripts/ select with $VQ_AMD_LIB.
#include "../../include/vq_amd.h"
#include "../../include/vq_amd_diag.h"
#include "vq_common.h"
#include "gemm_mfma.h"
#include "gemm_mfma256.h"
#include "gemm_mfma256p.h"
#include "gemm_mfma128x256p.h"
#include "encoder_kernels.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace vq { int require_init(); }
using namespace vq;

extern "C" {

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

// Diagnostic: the deep-prefetch mainloop as a 256 x 192 tile (timing ablation NARROW of gemm_tn256d_kernel: wrong results) beside
// the real 256 x 256 one.  narrow = 0 / 1; N counts 256-wide tile slots either way (a narrow run of N = 3072 stands for 12 tiles of 192
// columns = 2304).  Returns the average launch time, the clock held inside the K loop and the median K-loop cycles per workgroup.
int vq_debug_gemm_narrow(int M, int N, int K, int narrow, int reps, float* ms_avg, float* ghz_median, float* loop_cycles_median) {
    VQ_TRY(require_init());
    VQ_CHECK(ms_avg && ghz_median && loop_cycles_median && reps > 0 && M % 256 == 0 && N % 256 == 0 && K % 128 == 0, "vq_debug_gemm_narrow: bad argument");
    uint16_t *dA = nullptr, *dW = nullptr; float* dC = nullptr; unsigned long long* dS = nullptr;
    const int wgs = (M / 256) * (N / 256);
    VQ_HIP(hipMalloc(&dA, (size_t)M * K * 2)); VQ_HIP(hipMalloc(&dW, (size_t)N * K * 2)); VQ_HIP(hipMalloc(&dC, (size_t)M * N * 4));
    VQ_HIP(hipMalloc(&dS, (size_t)wgs * 16));
    std::vector<uint16_t> a16((size_t)M * K), w16((size_t)N * K);
    uint32_t r = 4242;
    for (auto& v : a16) { r = r * 1664525u + 1013904223u; v = __builtin_bit_cast(uint16_t, (_Float16)(((int)(r >> 8) % 2001 - 1000) * 1e-3f)); }
    for (auto& v : w16) { r = r * 1664525u + 1013904223u; v = __builtin_bit_cast(uint16_t, (_Float16)(((int)(r >> 8) % 2001 - 1000) * 1e-3f)); }
    VQ_HIP(hipMemcpy(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
    VQ_HIP(hipMemcpy(dW, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    typedef EpiStoreF32 E;
    VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256d_kernel<true, E, 1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
    VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn256d_kernel<true, E, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
    hipEvent_t e0, e1;
    VQ_HIP(hipEventCreate(&e0)); VQ_HIP(hipEventCreate(&e1));
    auto once = [&]() {
        if (narrow)
            hipLaunchKernelGGL((gemm_tn256d_kernel<true, E, 1, true, true>), dim3(wgs), dim3(G2_THREADS), G2_LDS_BYTES, nullptr, dA, K, dW, K, K, N / 256, E{dC, N}, 0, dS);
        else
            hipLaunchKernelGGL((gemm_tn256d_kernel<true, E, 1, true, false>), dim3(wgs), dim3(G2_THREADS), G2_LDS_BYTES, nullptr, dA, K, dW, K, K, N / 256, E{dC, N}, 0, dS);
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
    std::vector<float> ghz, cyc;
    for (int i = 0; i < wgs; ++i) if (st[2 * i + 1]) { ghz.push_back((float)st[2 * i] / (float)st[2 * i + 1] * 0.1f); cyc.push_back((float)st[2 * i]); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    *ghz_median = ghz.empty() ? 0.f : ghz[ghz.size() / 2];
    *loop_cycles_median = cyc.empty() ? 0.f : cyc[cyc.size() / 2];
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


// One GEMM kernel with one of the tower's epilogues, timed in isolation on fp16 operands (random, |v| <= 1).
//   kernel: 8 = 256x256 deep prefetch (gemm_tn256d), 24 = the hand-scheduled four-wave 256x256 (gemm_tn256a), 20 = persistent out-of-phase 128x256 (gemm_tn128x256p), 12 = the
//           non-persistent 128x256 two-per-CU experiment (EXPERIMENTS builds)
//   epi:    0 = fp32 store, 1 = bias + residual + 16-bit copy + LayerNorm row partials (out_proj / fc2), 2 = LayerNorm-consuming
//           quick-GELU 16-bit store (fc1), 3 = the same without GELU (qkv)
//   mode / dephase_cycles / grid: kernel 20 only (gemm_mfma128x256p.h); census: [grid][4] or null
int vq_debug_gemm_bench(int M, int N, int K, int kernel, int mode, int dephase_cycles, int epi, int reps, int grid,
                        float* ms_avg, unsigned long long* census) {
    VQ_TRY(require_init());
    VQ_CHECK(ms_avg && reps > 0 && epi >= 0 && epi <= 3, "vq_debug_gemm_bench: bad argument");
    uint16_t *dA = nullptr, *dW = nullptr, *dH = nullptr; float *dC = nullptr, *dV = nullptr; float2* dP = nullptr;
    unsigned long long* dS = nullptr;
    VQ_HIP(hipMalloc(&dA, (size_t)M * K * 2)); VQ_HIP(hipMalloc(&dW, (size_t)N * K * 2));
    VQ_HIP(hipMalloc(&dC, (size_t)M * N * 4)); VQ_HIP(hipMalloc(&dH, (size_t)M * N * 2));
    VQ_HIP(hipMalloc(&dV, (size_t)N * 4 * 2)); VQ_HIP(hipMalloc(&dP, (size_t)LN_MAX_GRANULES * M * 8));
    VQ_HIP(hipMalloc(&dS, (size_t)4096 * 4 * 8));        // (kernel 24's clock pairs: 8192 workgroups x 2)
    VQ_HIP(hipMemset(dC, 0, (size_t)M * N * 4)); VQ_HIP(hipMemset(dV, 0, (size_t)N * 8)); VQ_HIP(hipMemset(dP, 0, (size_t)LN_MAX_GRANULES * M * 8));
    VQ_HIP(hipMemset(dS, 0, (size_t)4096 * 4 * 8));
    {
        std::vector<uint16_t> a16((size_t)M * K), w16((size_t)N * K);
        uint32_t r = 2024;
        for (auto& v : a16) { r = r * 1664525u + 1013904223u; v = __builtin_bit_cast(uint16_t, (_Float16)(((int)(r >> 8) % 2001 - 1000) * 1e-3f)); }
        for (auto& v : w16) { r = r * 1664525u + 1013904223u; v = __builtin_bit_cast(uint16_t, (_Float16)(((int)(r >> 8) % 2001 - 1000) * 1e-3f)); }
        VQ_HIP(hipMemcpy(dA, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
        VQ_HIP(hipMemcpy(dW, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    }
    const LnPartials part{dP, (int64_t)M};
    auto with_epi = [&](auto&& fn) {
        switch (epi) {
            case 1: return fn(EpiBiasResidualLnF32<0, true>{dC, N, dV, dH, part});
            case 2: return fn(EpiLnH16<true, true>{dH, N, dV, dV + N, part, K / 64 > LN_MAX_GRANULES ? LN_MAX_GRANULES : K / 64, 1.0f / (float)K, 1e-5f});
            case 3: return fn(EpiLnH16<true, false>{dH, N, dV, dV + N, part, K / 64 > LN_MAX_GRANULES ? LN_MAX_GRANULES : K / 64, 1.0f / (float)K, 1e-5f});
            default: return fn(EpiStoreF32{dC, N});
        }
    };
    auto once = [&]() {
        return with_epi([&](auto e) -> int {
            if (kernel == 20) return launch_gemm_tn128x256p<true>(nullptr, dA, K, dW, K, M, N, K, e, mode, dephase_cycles, census ? dS : nullptr, grid);
#ifdef VQ_GEMM_EXPERIMENTS
            if (kernel == 12) return launch_gemm_tn128x256<true>(nullptr, dA, K, dW, K, M, N, K, e);
#endif
            if (kernel == 8) return launch_gemm_tn256d<true>(nullptr, dA, K, dW, K, M, N, K, e);
            if (kernel == 24) {              // mode: 0 = the product schedule, 1-5 = timing ablations (no DMA / no DMA, no reads / all waves in the same DMA slots (valid results) / no MFMAs / no MFMAs, no reads)
                if (mode == 1) return launch_gemm_tn256a<true, decltype(e), 1>(nullptr, dA, K, dW, K, M, N, K, e, census ? dS : nullptr);
                if (mode == 2) return launch_gemm_tn256a<true, decltype(e), 2>(nullptr, dA, K, dW, K, M, N, K, e, census ? dS : nullptr);
                if (mode == 3) return launch_gemm_tn256a<true, decltype(e), 3>(nullptr, dA, K, dW, K, M, N, K, e, census ? dS : nullptr);
                if (mode == 4) return launch_gemm_tn256a<true, decltype(e), 4>(nullptr, dA, K, dW, K, M, N, K, e, census ? dS : nullptr);
                if (mode == 5) return launch_gemm_tn256a<true, decltype(e), 5>(nullptr, dA, K, dW, K, M, N, K, e, census ? dS : nullptr);
                if (mode == 6) return launch_gemm_tn256a<true, decltype(e), 6>(nullptr, dA, K, dW, K, M, N, K, e, census ? dS : nullptr);
                if (mode == 7) return launch_gemm_tn256a<true, decltype(e), 7>(nullptr, dA, K, dW, K, M, N, K, e, census ? dS : nullptr);
                return launch_gemm_tn256a<true>(nullptr, dA, K, dW, K, M, N, K, e, census ? dS : nullptr);
            }
            return fail(VQ_ERR_INVALID, "vq_debug_gemm_bench: kernel %d is not available in this build", kernel);
        });
    };
    hipEvent_t e0, e1;
    VQ_HIP(hipEventCreate(&e0)); VQ_HIP(hipEventCreate(&e1));
    int rc = 0;
    for (int i = 0; i < 3 && rc == 0; ++i) rc = once();
    VQ_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < reps && rc == 0; ++i) rc = once();
    VQ_HIP(hipEventRecord(e1, nullptr));
    VQ_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    VQ_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / reps;
    if (rc == 0 && census && kernel == 24) {       // census[0] = median clock inside the K loop (MHz), census[1] = median K-loop cycles per workgroup
        const int wgs = (M / 256) * (N / 256) > 8192 ? 8192 : (M / 256) * (N / 256);
        std::vector<unsigned long long> st((size_t)wgs * 2);
        VQ_HIP(hipMemcpy(st.data(), dS, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> mhz; std::vector<unsigned long long> cyc;
        for (int i = 0; i < wgs; ++i) if (st[2 * i + 1]) { mhz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0); cyc.push_back(st[2 * i]); }
        std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
        census[0] = mhz.empty() ? 0 : (unsigned long long)mhz[mhz.size() / 2];
        census[1] = cyc.empty() ? 0 : cyc[cyc.size() / 2];
    } else
    if (rc == 0 && census) {
        const int g = grid > 0 ? grid : 512;
        VQ_HIP(hipMemcpy(census, dS, (size_t)(g > 4096 ? 4096 : g) * 4 * 8, hipMemcpyDeviceToHost));
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC); (void)hipFree(dH); (void)hipFree(dV); (void)hipFree(dP); (void)hipFree(dS);
    return rc;
}

}  // extern "C"

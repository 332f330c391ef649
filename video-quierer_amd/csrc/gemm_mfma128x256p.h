// 128(m) x 256(n) bf16/fp16 TN GEMM, TWO persistent workgroups per CU, run OUT OF PHASE.
//
// Why.  A 256x256 workgroup (gemm_mfma256d.h) owns its CU — 130 KiB of LDS and every register — so the matrix pipes idle
// while it waits for its first operands and while it pushes its epilogue out: a third of all GEMM workgroup-cycles of the
// tower (tower stamps).  Two half-size workgroups per CU can cover each other's prologue / epilogue with MFMAs
// — but only if they are not in the same phase of their lives, and two workgroups of one launch that land on a CU together
// STAY together: they fill, multiply and drain in step (gemm_mfma128x256.h measured exactly that: no gain).  Their phase
// difference is marginally stable — whatever lag they start with, they keep (while both multiply they share the pipes; the
// one ahead pulls away alone by as much as the one behind catches up when it is alone) — so the lag has to be put in once:
//
//   * the grid is persistent: at most 2 workgroups per CU, each walking tiles  c, c + G, c + 2G ...  (c = its XCD-contiguous
//     index), so a CU keeps the same pair for the whole launch and pays no dispatch between tiles;
//   * the workgroup that holds the SECOND wave slot of its SIMDs (HW_ID.WAVE_ID odd) waits `dephase_cycles` before its
//     first tile — about half a tile period, chosen by the host from K and the epilogue class.  Speed only: results do not
//     depend on which workgroups wait, nor on whether the hardware id means what it is taken to mean.
//
// Mainloop, LDS image, hazards: those of gemm_tn128x256_kernel (one wave per SIMD and workgroup, 128 x 64 per wave, 32-wide
// K sub-tiles through a 3-slot 72 KiB ring, register double-buffered fragments, one barrier per sub-tile, buffer_load ... lds).
// Between tiles: every wave's epilogue strip (ring memory) and the row statistics are dead before the next tile's first
// LDS-DMA / statistics write — one barrier behind the epilogue.
// Requirements: M % 128 == 0, N % 256 == 0, K % 64 == 0, K >= 128, lda/ldw % 8 == 0.
#pragma once
#include "vq_common.h"
#include "gemm_mfma.h"
#include <cstdlib>

namespace vq {

constexpr int GP_BM = 128, GP_BN = 256, GP_SUB_K = 32, GP_THREADS = 256;
constexpr int GP_A_BYTES = GP_BM * GP_SUB_K * 2;            // 8 KiB
constexpr int GP_SLOT = (GP_BM + GP_BN) * GP_SUB_K * 2;     // 24 KiB
constexpr int GP_NSLOT = 3;
constexpr int GP_LDS_BYTES = GP_NSLOT * GP_SLOT;            // 72 KiB
constexpr int GP_ROWSTAT_BYTES = GP_BM * 8;

// mode bits: 1 = the odd-slot workgroup waits dephase_cycles before its first tile; 2 = the odd-slot workgroup runs its
// MFMA clusters at priority 0/1 and the even one at 2/3 (no wait: the even workgroup behaves as if alone on the CU, the odd
// one takes the pipe cycles it leaves); 4 = census: lane 0 of wave 0 records {HW_ID, XCC_ID, s_memtime at start, at end}
template <bool IS_F16, class Epi>
__global__ __launch_bounds__(GP_THREADS, 2)
void gemm_tn128x256p_kernel(const uint16_t* __restrict__ A, int lda,
                            const uint16_t* __restrict__ W, int ldw,
                            int K, int tiles_m, int tiles_n, Epi epi, int mode, int dephase_cycles,
                            unsigned long long* __restrict__ census) {
    typedef mfma_op<IS_F16> op;
    typedef typename op::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // HW_REG_HW_ID (4): WAVE_ID [3:0] = the wave's slot on its SIMD.  With two 4-wave workgroups per CU the first holds
    // slot 0 of every SIMD and the second slot 1.
    const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const bool odd_slot = (hw_id & 1u) != 0;
    unsigned long long t_start = 0;
    if (mode & 4) t_start = __builtin_amdgcn_s_memtime();
    if ((mode & 1) && odd_slot && dephase_cycles > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)dephase_cycles) __builtin_amdgcn_s_sleep(32);
    }
    const bool low_prio = (mode & 2) && odd_slot;
    const bool high_prio = (mode & 2) && !odd_slot;

    const int nwg = (int)gridDim.x;
    const int c0 = xcd_remap(blockIdx.x, nwg);
    const int tiles = tiles_m * tiles_n;

    // LDS-DMA: a 1-KiB piece = 16 rows x 64 B.  Wave w fills A pieces 2w, 2w+1 (rows 32w..32w+31) and W pieces 4w..4w+3
    // (rows 64w..64w+63: its own columns).  One lane offset per operand; piece row offsets and the K offset are scalar.
    const int srow = lane >> 2;
    const int schunk = (lane & 3) ^ (((lane >> 5) & 1) * 2);      // logical chunk stored at physical slot lane&3
    const int a_v = ((wave * 32 + srow) * lda + schunk * 8) * 2;
    const int w_v = ((wave * 64 + srow) * ldw + schunk * 8) * 2;
    const int a_p16 = 16 * lda * 2, w_p16 = 16 * ldw * 2;         // 16 source rows, bytes
    const int a_dst = wave * 2048, w_dst = GP_A_BYTES + wave * 4096;

    const int frow = lane & 15, fgrp = lane >> 4;
    const int pchunk = fgrp ^ (((frow >> 3) & 1) * 2);
    const int a_base = frow * 64 + pchunk * 16;                              // + mi*1024
    const int w_base = GP_A_BYTES + (wave * 64 + frow) * 64 + pchunk * 16;   // + ni*1024

    const int nsub = K / GP_SUB_K;       // even, >= 4
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto prio_up = [&]() __attribute__((always_inline)) {
        if (high_prio) __builtin_amdgcn_s_setprio(3);
        else if (low_prio) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(1);
    };
    auto prio_down = [&]() __attribute__((always_inline)) {
        if (high_prio) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);
    };

    for (int t = c0; t < tiles; t += nwg) {
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        const int m0 = tm * GP_BM, n0 = tn * GP_BN;
        const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, 0x7fffffff, 0x00020000);

        auto stage = [&](int slot, int sub) __attribute__((always_inline)) {
            char* base = smem + slot * GP_SLOT;
            const int koff = __builtin_amdgcn_readfirstlane(sub * (GP_SUB_K * 2));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + a_dst), 16, a_v, koff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_a, (lds_void_t*)(base + a_dst + 1024), 16, a_v, koff + a_p16, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst), 16, w_v, koff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst + 1024), 16, w_v, koff + w_p16, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst + 2048), 16, w_v, koff + 2 * w_p16, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (lds_void_t*)(base + w_dst + 3072), 16, w_v, koff + 3 * w_p16, 0, 0);
        };

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        frag af0[8], wf0[4], af1[8], wf1[4];

#define VQ_READ_FRAGS(AF, WF, SLOT)                                                                       \
    do {                                                                                                  \
        const char* b__ = smem + (SLOT) * GP_SLOT;                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) AF[i] = *(const frag*)(b__ + a_base + i * 1024);    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) WF[j] = *(const frag*)(b__ + w_base + j * 1024);    \
    } while (0)
#define VQ_MFMA_ROWS(AF, WF, I0, I1)                                                                      \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = I0; i < I1; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = op::run(WF[j], AF[i], acc[i][j]);   \
    } while (0)

        // prologue: sub-tiles 0, 1, 2 in flight; 0 landed -> its fragments into set 0; 1 landed
        stage(0, 0); stage(1, 1); stage(2, 2);
        const Epi epi_wg = epi_bind_rowstats<GP_BM>(epi, (float2*)(smem + GP_LDS_BYTES), m0, tid, GP_THREADS);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        barrier();
        VQ_READ_FRAGS(af0, wf0, 0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        barrier();

        int s = 0;                                  // slot of sub-tile p
        for (int p = 0; p < nsub; p += 2) {
            const int s1 = s == 2 ? 0 : s + 1, s2 = s1 == 2 ? 0 : s1 + 1;       // slots of p+1, p+2
            // ---- sub-tile p on set 0; prefetch the fragments of p+1 into set 1; refill slot s (= slot of p+3) ----
            prio_up();
            VQ_MFMA_ROWS(af0, wf0, 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            VQ_READ_FRAGS(af1, wf1, s1);                                        // p+1 < nsub always (nsub even)
            if (p + 3 < nsub) stage(s, p + 3);
            __builtin_amdgcn_sched_barrier(0);
            VQ_MFMA_ROWS(af0, wf0, 2, 8);
            prio_down();
            if (p + 3 < nsub)      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // p+2 landed, p+3 in flight
            else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            barrier();
            // ---- sub-tile p+1 on set 1; prefetch p+2 into set 0; refill slot s1 (= slot of p+4) ----
            prio_up();
            VQ_MFMA_ROWS(af1, wf1, 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            if (p + 2 < nsub) VQ_READ_FRAGS(af0, wf0, s2);
            if (p + 4 < nsub) stage(s1, p + 4);
            __builtin_amdgcn_sched_barrier(0);
            VQ_MFMA_ROWS(af1, wf1, 2, 8);
            prio_down();
            if (p + 4 < nsub)      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // p+3 landed, p+4 in flight
            else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            barrier();
            s = s2;
        }
#undef VQ_READ_FRAGS
#undef VQ_MFMA_ROWS
        // the last barrier ended the last iteration: every fragment read has returned, LDS is free for the epilogue strips
        wave_epilogue<8>(smem + wave * EPI_WAVE_BYTES, acc, m0, n0 + wave * 64, lane, epi_wg);
        if (t + nwg < tiles) {
            // strips and row statistics are read by their owners only until here; the next tile's DMA / statistics overwrite them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            barrier();
        }
    }
    if ((mode & 4) && census && tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);       // HW_REG_XCC_ID [3:0]
        unsigned long long* d = census + (size_t)blockIdx.x * 4;
        d[0] = hw_id; d[1] = xcc; d[2] = t_start; d[3] = __builtin_amdgcn_s_memtime();
    }
}

// Workgroups of the persistent grid: two per CU at most, never more than there are tiles.
static inline int gp_grid(int tiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return tiles < 2 * cus ? tiles : 2 * cus;
}

// Lag of the second workgroup of a CU: about half of what one tile costs a workgroup that shares its CU (K loop ~1.5k cycles
// per 32-wide sub-tile with both multiplying, plus prologue and epilogue).  Speed only.
static inline int gp_dephase_cycles(int K) {
    static int v = -2;
    if (v == -2) { const char* e = getenv("VQ_AMD_GP_DEPHASE"); v = e ? atoi(e) : -1; }
    if (v >= 0) return v;
    return (K / GP_SUB_K) * 750 + 6000;
}

template <bool IS_F16, class Epi>
static int launch_gemm_tn128x256p(hipStream_t st, const uint16_t* A, int lda, const uint16_t* W, int ldw,
                                  int M, int N, int K, const Epi& epi, int mode, int dephase_cycles,
                                  unsigned long long* census = nullptr, int grid_override = 0) {
    VQ_CHECK(M > 0 && M % GP_BM == 0 && N % GP_BN == 0 && K % (2 * GP_SUB_K) == 0 && K >= 4 * GP_SUB_K,
             "gemm_tn128x256p: shape M=%d N=%d K=%d is not tile-aligned (128/256/64)", M, N, K);
    VQ_CHECK(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_tn128x256p: operands must be 16-byte aligned with lda/ldw %% 8 == 0");
    static bool attr_set = false;
    if (!attr_set) {
        VQ_HIP(hipFuncSetAttribute((const void*)gemm_tn128x256p_kernel<IS_F16, Epi>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, GP_LDS_BYTES + GP_ROWSTAT_BYTES));
        attr_set = true;
    }
    const int tiles_m = M / GP_BM, tiles_n = N / GP_BN;
    const int grid = grid_override > 0 ? grid_override : gp_grid(tiles_m * tiles_n);
    hipLaunchKernelGGL((gemm_tn128x256p_kernel<IS_F16, Epi>), dim3(grid), dim3(GP_THREADS),
                       GP_LDS_BYTES + (epi_row_in<Epi>::value ? GP_ROWSTAT_BYTES : 0), st, A, lda, W, ldw, K, tiles_m, tiles_n, epi,
                       mode, dephase_cycles, census);
    VQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace vq

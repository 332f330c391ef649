/* Diagnostic entry points of libvq_amd — NOT part of the product library.
 *
 * Built only by `make -C video-quierer_amd/csrc DIAG=1 OUT=../lib/libvq_amd_diag.so OBJDIR=../lib/obj_diag`
 * (EXPERIMENTS=1 and STAMPS=1 imply DIAG=1); scripts/ select that build with $VQ_AMD_LIB.  Results of the ablation
 * entry points are numerically invalid by design: they time a mainloop with parts removed.
 */
#ifndef VQ_AMD_DIAG_H
#define VQ_AMD_DIAG_H
#include "vq_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic build of the 256x256 mainloop with in-kernel s_memtime stamps: workgroup 0, 8 waves x
 * 768 stamps (3 per phase: start of read half, after the first barrier, after the MFMAs).
 * diag: bit0 skip the in-loop DMA, bit1 skip the ds_reads, bit2 skip the MFMAs (timing ablations; results invalid). */
int vq_debug_gemm_stamps(int M, int N, int K, int diag, unsigned long long* stamps);

/* Average time of one GEMM mainloop with parts removed (results invalid).  kernel: 1 = 128x128, 2 = 256x256
 * four-phase, 3 = 256x256 ring, 8 = deep prefetch ...; diag: bit0 no in-loop DMA, bit1 no ds_reads, bit2 no MFMAs,
 * bit3 no barriers (ring only). */
int vq_debug_gemm_ablate(int M, int N, int K, int kernel, int diag, int reps, float* ms_avg);

/* Average launch time of the deep-prefetch 256x256 mainloop on random data and the clock (GHz) the chip holds inside
 * its K loop (d s_memtime / d s_memrealtime, median over workgroups). */
int vq_debug_gemm_clock(int M, int N, int K, int reps, float* ms_avg, float* ghz_median);
/* The deep-prefetch mainloop as a 256 x 192 tile (a timing ablation: results are wrong) beside the 256 x 256 one: what a
 * q|k|v-of-one-head tile would cost.  N counts 256-wide tile slots in both forms. */
int vq_debug_gemm_narrow(int M, int N, int K, int narrow, int reps, float* ms_avg, float* ghz_median, float* loop_cycles_median);

/* s_memtime stamps of workgroup 0 of that mainloop, four per phase (phase start, before the mid barrier, before the
 * MFMAs, after them): stamps[8 waves][512]. */
int vq_debug_gemm_stamps_deep(int M, int N, int K, int reps, unsigned long long* stamps);

/* One GEMM kernel with one of the tower's epilogues, timed in isolation (fp16 operands, random data).
 * kernel: 8 = 256x256 deep prefetch, 24 = the four-wave 256x256 kernel whose K loop is one hand-scheduled asm text
 * (csrc/gemm_asm256.h; mode 0 = the product schedule, 1-5 = timing ablations of csrc/gemm_asm256_loop.inc; census: null or
 * >= 2 entries that receive the median clock inside the K loop in MHz and the median K-loop cycles per workgroup), 20 = persistent out-of-phase 128x256 (two workgroups per CU), 12 = the
 * non-persistent 128x256 experiment (EXPERIMENTS builds).  epi: 0 = fp32 store, 1 = bias + residual + 16-bit copy +
 * LayerNorm row partials, 2 = LayerNorm-consuming quick-GELU 16-bit store, 3 = the same without GELU.
 * mode / dephase_cycles / grid: kernel 20 only (csrc/gemm_mfma128x256p.h).  census: null or [grid][4] =
 * {HW_ID, XCC_ID, s_memtime at start, at end} per workgroup of the LAST launch. */
int vq_debug_gemm_bench(int M, int N, int K, int kernel, int mode, int dephase_cycles, int epi, int reps, int grid,
                        float* ms_avg, unsigned long long* census);

#ifdef __cplusplus
}
#endif
#endif

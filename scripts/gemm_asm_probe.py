#!/usr/bin/env python3
"""A/B in ONE process, interleaved rounds: the 256x256 deep-prefetch kernel (8) against the hand-scheduled four-wave kernel
(24, gemm_asm256.h) with the tower's own epilogues on the tower's shapes, and on large GEMMs.  Needs a DIAG build:
    make -C video-quierer_amd/csrc DIAG=1 OUT=../lib/libvq_amd_diag.so OBJDIR=../lib/obj_diag
    VQ_AMD_LIB=video-quierer_amd/lib/libvq_amd_diag.so python scripts/gemm_asm_probe.py"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
_lib.init(0)
lib = _lib.load()


def run(m, n, k, kernel, epi, reps=20, mode=0, clock=None):
    ms = ctypes.c_float(0)
    _lib.check(lib.vq_debug_gemm_bench(m, n, k, kernel, mode, 0, epi, reps, 0, ctypes.byref(ms), clock.ctypes.data_as(ctypes.c_void_p) if clock is not None else None))
    return ms.value


def clocked(m, n, k, epi, mode):
    """(us, MHz inside the K loop, cycles per K-tile) of kernel 24 in `mode`"""
    c = np.zeros(2048 * 4, dtype=np.uint64)
    t = run(m, n, k, 24, epi, reps=10, mode=mode, clock=c)
    return t * 1e3, int(c[0]), int(c[1]) / (k // 64)


from video_quierer_amd.encoder import debug_gemm
rng = np.random.default_rng(5)
for (m, n, k) in [(256, 256, 128), (256, 256, 256), (512, 256, 384), (1280, 768, 768), (2560, 2304, 3072)]:
    a = rng.integers(-4, 5, (m, k)).astype(np.float32)
    w = rng.integers(-4, 5, (n, k)).astype(np.float32)
    ref = a @ w.T
    for f16 in (True, False):
        got = debug_gemm(a, w, use_f16=f16, kernel=24)
        bad = np.argwhere(got != ref)
        assert bad.size == 0, f"kernel 24 wrong at {(m, n, k)} f16={f16}: {len(bad)} elements, first {bad[:5].tolist()}, max err {np.abs(got - ref).max()}"
    print(f"kernel 24 exact on integer operands at {(m, n, k)}", flush=True)

cases = [("fc1", 12800, 3072, 768, 2), ("qkv", 12800, 2304, 768, 3), ("out", 12800, 768, 768, 1), ("fc2", 12800, 768, 3072, 1),
         ("fc1 store", 12800, 3072, 768, 0), ("fc2 store", 12800, 768, 3072, 0),
         ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0), ("16384x4096x4096", 16384, 4096, 4096, 0)]
if len(sys.argv) > 1:
    cases = [c for c in cases if any(a in c[0] for a in sys.argv[1:])]
for name, m, n, k, epi in cases:
    t = {8: [], 24: []}
    for _ in range(4):
        for kern in (8, 24):
            t[kern].append(run(m, n, k, kern, epi))
    a, b = np.median(t[8]) * 1e3, np.median(t[24]) * 1e3
    fl = 2.0 * m * n * k
    print(f"{name:16s} M={m} N={n} K={k} epi={epi}: kernel 8 {a:8.1f} us ({fl / a / 1e6:6.0f} TFLOP/s)   kernel 24 {b:8.1f} us ({fl / b / 1e6:6.0f} TFLOP/s)   ratio {a / b:.3f}", flush=True)
    if os.environ.get("VQ_PROBE_CLOCK"):
        for mode, what in ((0, "product schedule"), (1, "no DMA"), (2, "no DMA, no fragment reads"), (4, "no MFMAs")):
            if epi != 0 and mode != 0:
                continue
            us, mhz, cpt = clocked(m, n, k, epi, mode)
            print(f"{'':16s} kernel 24 {what}: {us:.1f} us, {mhz} MHz inside the K loop, {cpt:.0f} cycles per K-tile (= {cpt / mhz:.3f} us)", flush=True)
    if os.environ.get("VQ_PROBE_ABLATE") and epi == 0:
        ab = {v: np.median([run(m, n, k, 24, epi, mode=v) for _ in range(3)]) * 1e3 for v in (1, 2, 3, 4, 5, 6, 7)}
        t0 = np.median([run(m, n, k, 24, epi, mode=0) for _ in range(3)]) * 1e3
        print(f"{'':16s} the batch scan's top-2 fold in this four-wave form (a quarter of a row tile's fold per two K-tiles, results invalid): "
              f"product loop {t0:.1f} us, + fold as a burst behind the accumulators' last MFMAs {ab[6]:.1f} ({ab[6] / t0:.3f}x), "
              f"+ fold spread one instruction per MFMA {ab[7]:.1f} ({ab[7] / t0:.3f}x)", flush=True)
        print(f"{'':16s} ablations of kernel 24 (results invalid): no DMA {ab[1]:.1f} us ({fl / ab[1] / 1e6:.0f}), no DMA + no fragment reads {ab[2]:.1f} ({fl / ab[2] / 1e6:.0f}), every wave in the same DMA slots {ab[3]:.1f} ({fl / ab[3] / 1e6:.0f}), no MFMAs {ab[4]:.1f}, no MFMAs + no fragment reads {ab[5]:.1f}", flush=True)

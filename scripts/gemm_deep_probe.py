#!/usr/bin/env python3
"""A/B of the 256x256 mainloops in ONE process, interleaved rounds (cdna_hip_programming.md rule 24):
kernel 2 = four-phase (prefetch 1-3 phases ahead), 8 = four-phase with the deep prefetch (5-6 phases ahead),
10 = 32-wide ring with register double-buffered fragments and one barrier per sub-tile, 5 = 160x256 ring.  Plain fp32-store epilogue, random data.  Then the clock held inside kernel 8's K loop."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
_lib.init(0)
lib = _lib.load()
def run(m, n, k, kernel, reps=20):
    ms = ctypes.c_float(0)
    _lib.check(lib.vq_debug_gemm_ablate(m, n, k, kernel, 0, reps, ctypes.byref(ms)))
    return ms.value
shapes = [(8192, 8192, 8192), (16384, 4096, 4096), (12800, 3072, 768), (12800, 2304, 768), (12800, 768, 3072), (12800, 768, 768)]
for (m, n, k) in shapes:
    res = {8: [], int(os.environ.get("PROBE_KERNEL", "11")): []}
    if m % 160 == 0 and n == 768:
        res[5] = []
    for rnd in range(5):
        for kern in res:
            res[kern].append(run(m, n, k, kern, 10 if m * n * k > 2e11 else 20))
    fl = 2.0 * m * n * k
    print(f"M={m} N={n} K={k}: " + "  ".join(f"kernel {kern}: median {np.median(v)*1e3:8.1f} us min {min(v)*1e3:8.1f} us = {fl/np.median(v)/1e9:6.0f} TFLOP/s"
                                           for kern, v in res.items()), flush=True)
for (m, n, k) in [(16384, 4096, 4096), (12800, 3072, 768), (12800, 768, 3072)]:
    ms, ghz = ctypes.c_float(0), ctypes.c_float(0)
    _lib.check(lib.vq_debug_gemm_clock(m, n, k, 200, ctypes.byref(ms), ctypes.byref(ghz)))
    print(f"clock inside kernel 8's K loop, M={m} N={n} K={k}: {ghz.value:.3f} GHz, launch {ms.value*1e3:.1f} us "
          f"({2.0*m*n*k/ms.value/1e9:.0f} TFLOP/s; peak at this clock {2.5e3*ghz.value/2.4:.0f} TFLOP/s)", flush=True)

#!/usr/bin/env python3
"""This build's mainloops on a large square problem (steady state, many rounds): kernel 2 = four-phase 256x256,
3 = ring, 1 = 128x128.  Calibration against scripts/vendor_gemm_probe.py's `square 8192` line."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
_lib.init(0)
lib = _lib.load()
def run(m, n, k, kernel):
    ms = ctypes.c_float(0)
    _lib.check(lib.vq_debug_gemm_ablate(m, n, k, kernel, 0, 20, ctypes.byref(ms)))
    return ms.value
for (m, n, k) in ((8192, 8192, 8192), (16384, 4096, 4096), (12800, 3072, 768), (12800, 2304, 768), (12800, 768, 3072), (12800, 768, 768)):
    for kernel in (2, 7):
        t = run(m, n, k, kernel)
        print(f"M={m} N={n} K={k} kernel {kernel}: {t*1e3:9.1f} us  {2.0*m*n*k/t/1e9:7.0f} TFLOP/s", flush=True)

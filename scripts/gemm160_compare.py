#!/usr/bin/env python3
"""Times the 256x256 four-phase kernel against the 160x256 ring kernel on the N=768 shapes of the B/32 tower."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
_lib.init(0)
lib = _lib.load()
def run(m, n, k, kernel):
    ms = ctypes.c_float(0)
    _lib.check(lib.vq_debug_gemm_ablate(m, n, k, kernel, 0, 50, ctypes.byref(ms)))
    return ms.value
for (m, n, k) in ((12800, 768, 3072), (12800, 768, 768), (12800, 2304, 768), (12800, 3072, 768), (3200, 2304, 768), (3200, 768, 3072)):
    fl = 2.0 * m * n * k
    for kernel in (2, 3, 5):
        if kernel != 5 and m % 256:
            continue
        t = run(m, n, k, kernel)
        print(f"M={m} N={n} K={k} kernel {kernel}: {t*1e3:8.1f} us  {fl/t/1e9:7.0f} TFLOP/s", flush=True)

#!/usr/bin/env python3
"""Times the GEMM mainloops with parts removed (diagnostic; invalid results).  usage: gemm_ablate.py M N K"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
m, n, k = (int(v) for v in sys.argv[1:4])
_lib.init(0)
lib = _lib.load()
def run(kernel, diag):
    ms = ctypes.c_float(0)
    _lib.check(lib.vq_debug_gemm_ablate(m, n, k, kernel, diag, 20, ctypes.byref(ms)))
    return ms.value
fl = 2.0 * m * n * k
names = {0: "full", 1: "-dma", 2: "-reads", 4: "-mfma", 3: "-dma-reads", 5: "-dma-mfma", 6: "-reads-mfma", 7: "-all", 8: "-barriers", 15: "-all-barriers", 12: "-mfma-barriers", 11: "-dma-reads-barriers (mfma only)"}
names[16] = "ring5 full"; names[17] = "ring5 -dma"; names[20] = "ring5 -mfma"
for kernel in (1, 2, 3, 4):
    for diag in ([0] if kernel in (1, 4) else [0, 1, 4] + ([16, 17, 20] if kernel == 3 else [])):
        t = run(kernel, diag)
        print(f"kernel {kernel} {names[diag]:32s} {t*1e3:8.1f} us  {fl/t/1e9:7.0f} TF-equivalent")

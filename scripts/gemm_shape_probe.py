#!/usr/bin/env python3
"""What a K-tile of the 256x256 deep-prefetch mainloop costs as the GEMM's shape changes (plain fp32-store epilogue, fp16
random operands): workgroup rounds x K-tiles -> microseconds per K-tile per workgroup.  Needs a DIAG build ($VQ_AMD_LIB)."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
_lib.init(0)
lib = _lib.load()
def run(m, n, k, reps=10):
    ms = ctypes.c_float(0)
    _lib.check(lib.vq_debug_gemm_bench(m, n, k, 8, 0, 0, 0, reps, 0, ctypes.byref(ms), None))
    return ms.value
shapes = [(12800, 3072, 768), (12800, 3072, 1536), (12800, 3072, 3072), (12800, 3072, 6144), (51200, 3072, 768), (12800, 12288, 768),
          (16384, 4096, 4096), (8192, 8192, 8192), (65536, 1024, 512), (16384, 16384, 512), (65536, 4096, 512)]
for (m, n, k) in shapes:
    t = min(run(m, n, k) for _ in range(3))
    tiles = (m // 256) * (n // 256)
    rounds = tiles / 256.0
    kt = k // 64
    per = t * 1e3 / (np.ceil(rounds) * kt)
    print(f"M={m:6d} N={n:6d} K={k:5d}: {t*1e3:9.1f} us  {2.0*m*n*k/t/1e9:6.0f} TFLOP/s  tiles {tiles} = {rounds:.2f} rounds x {kt} K-tiles -> "
          f"{per:.3f} us per K-tile and round (epilogue / prologue included)", flush=True)

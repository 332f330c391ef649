#!/usr/bin/env python3
"""Prices a 256 x 192 tile in the deep-prefetch mainloop ("go only if an isolated probe of the 192-wide tile loses
< 5 % against the 256-wide one"): the q|k|v GEMM of the tower as 450 tiles of 256 x 256 (today) and as 624 tiles of 256 x 192 —
one head's q | k | v for five images (250-row image-aligned tiles: 52 tile rows) — the latter as the NARROW timing ablation of
gemm_tn256d_kernel (a wave's second W sub-block is one MFMA column tile instead of two: 48 MFMAs, 20 fragment reads, 14 LDS-DMA
pieces per K-tile and wave; results wrong).  Needs a DIAG build ($VQ_AMD_LIB)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
_lib.init(0)
lib = _lib.load()


def run(m, n, k, narrow, reps=20):
    ms, ghz, cyc = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_float(0)
    _lib.check(lib.vq_debug_gemm_narrow(m, n, k, narrow, reps, ctypes.byref(ms), ctypes.byref(ghz), ctypes.byref(cyc)))
    return ms.value * 1e3, ghz.value, cyc.value


K = 768
for name, (m_w, n_w), (m_n, n_n) in (("q|k|v as today (50 x 9 tiles of 256 x 256) / as 52 x 12 tiles of 256 x 192", (12800, 2304), (13312, 3072)),
                                     ("same row count both ways (50 tile rows): 450 wide / 600 narrow tiles", (12800, 2304), (12800, 3072)),
                                     ("one full round of 256 workgroups each", (16384, 1024), (16384, 1024))):
    res = {}
    for rnd in range(3):
        for tag, (m, n, narrow) in (("256 x 256", (m_w, n_w, 0)), ("256 x 192", (m_n, n_n, 1))):
            res.setdefault(tag, []).append(run(m, n, K, narrow))
    w = np.median(np.array(res["256 x 256"]), axis=0); nr = np.median(np.array(res["256 x 192"]), axis=0)
    print(f"{name}:\n    256 x 256: {w[0]:7.1f} us, {w[1]:.2f} GHz in the K loop, {w[2] / (K // 64):6.0f} cycles per K-tile\n"
          f"    256 x 192: {nr[0]:7.1f} us, {nr[1]:.2f} GHz, {nr[2] / (K // 64):6.0f} cycles per K-tile ({nr[2] / w[2]:.3f} of the wide tile's for 0.75 of its MFMAs)"
          f"   -> whole GEMM {nr[0] / w[0]:.3f}x (fp32-store epilogue of 256 columns in both: the narrow form's real epilogue is 25 % smaller)", flush=True)

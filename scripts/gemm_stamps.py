#!/usr/bin/env python3
"""Per-phase timeline of the 256x256 GEMM mainloop from in-kernel s_memtime stamps (diagnostic build).
usage: gemm_stamps.py M N K"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
m, n, k = (int(v) for v in sys.argv[1:4])
diag = int(sys.argv[4]) if len(sys.argv) > 4 else 0
_lib.init(0)
st = np.zeros((8, 768), dtype=np.uint64)
_lib.check(_lib.load().vq_debug_gemm_stamps(m, n, k, diag, st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
st = st.astype(np.int64)
nph = min(768 // 3, (k // 64) * 4)
t0 = st[:, 0].min()
rel = st - t0
for w in (0, 1, 4, 5):
    s = rel[w, :nph * 3].reshape(nph, 3)
    read_half = s[:, 1] - s[:, 0]          # reads + DMA issue + first barrier wait
    mfma_half = s[:, 2] - s[:, 1]          # lgkmcnt + 16 MFMAs
    close = np.append(s[1:, 0] - s[:-1, 2], 0)   # closing barrier wait
    print(f"wave {w}: start {s[0,0]}  phase period mean {np.diff(s[:,0]).mean():.0f} cycles")
    for name, arr in (("read_half", read_half), ("mfma_half", mfma_half), ("close_barrier", close)):
        a = arr[4:nph - 4]
        byp = [a[i::4].mean() for i in range(4)]
        print(f"   {name:14s} mean {a.mean():6.0f}  by phase-of-tile {[round(x) for x in byp]}")
print("diag", diag, "first 4 phases wave0 vs wave4 (S0,S1,S2):")
for p in range(4):
    print(p, rel[0, 3*p:3*p+3].tolist(), rel[4, 3*p:3*p+3].tolist())

#!/usr/bin/env python3
"""Per-phase timeline of the deep-prefetch 256x256 mainloop from in-kernel s_memtime stamps (diagnostic build):
four stamps per phase — S0 phase start, S1 before the mid barrier (reads + DMA issued, counted vmcnt done), S2 after
the barrier and lgkmcnt(0), S3 after the 16 MFMAs.  usage: gemm_stamps_deep.py M N K"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
m, n, k = (int(v) for v in sys.argv[1:4])
_lib.init(0)
st = np.zeros((8, 512), dtype=np.uint64)
_lib.check(_lib.load().vq_debug_gemm_stamps_deep(m, n, k, 20, st.ctypes.data_as(ctypes.c_void_p)))
st = st.astype(np.int64)
nph = min(512 // 4, (k // 64) * 4)
for w in (0, 1, 4, 5):
    s = st[w, :nph * 4].reshape(nph, 4)
    own = s[:, 1] - s[:, 0]            # this wave's read-half work (ds_read issue, DMA issue, counted vmcnt)
    wait_mid = s[:, 2] - s[:, 1]       # mid barrier + lgkmcnt(0)
    mfma = s[:, 3] - s[:, 2]           # 16 MFMAs
    wait_end = np.append(s[1:, 0] - s[:-1, 3], 0)   # end barrier
    period = np.diff(s[:, 0]).mean()
    print(f"wave {w}: phase period {period:.0f} cycles (ideal 2 x 256 of MFMA per SIMD)")
    for name, arr in (("read-half work", own), ("mid barrier + lgkm", wait_mid), ("16 MFMAs", mfma), ("end barrier", wait_end)):
        a = arr[8:nph - 8]
        print(f"   {name:20s} mean {a.mean():6.0f}   by phase of tile {[round(float(a[i::4].mean())) for i in range(4)]}")

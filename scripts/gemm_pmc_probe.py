#!/usr/bin/env python3
"""Workload for `rocprofv3 --pmc ...`: the 256x256 mainloops (kernel 2 four-phase, 8 deep prefetch) on a large shape and
on the tower's fc1 / fc2 shapes, a few launches each (plain fp32-store epilogue, random data)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
_lib.init(0)
lib = _lib.load()
for (m, n, k) in ((16384, 4096, 4096), (12800, 3072, 768), (12800, 768, 3072)):
    for kern in (2, 8):
        ms = ctypes.c_float(0)
        _lib.check(lib.vq_debug_gemm_ablate(m, n, k, kern, 0, 4, ctypes.byref(ms)))
        print(m, n, k, kern, ms.value, flush=True)

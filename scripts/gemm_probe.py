#!/usr/bin/env python3
"""Runs the production GEMM mainloops on one shape through vq_debug_gemm (for rocprofv3 --pmc / --kernel-trace).
usage: gemm_probe.py M N K kernel [reps]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.encoder import debug_gemm

m, n, k, kern = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rng = np.random.default_rng(0)
a = rng.standard_normal((m, k)).astype(np.float32)
w = rng.standard_normal((n, k)).astype(np.float32)
for _ in range(reps):
    t = time.time(); c = debug_gemm(a, w, kernel=kern); dt = time.time() - t
print("done", c.shape, float(np.abs(c).mean()))

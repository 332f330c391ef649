#!/usr/bin/env python3
"""Where a synchronous `index.search(query, k)` (the reference caller's call, video_search_system.py:297) spends its time:
p50 of (a) the Python drop-in call, (b) the raw C entry point vq_index_search through ctypes on pre-made arrays, (c) the
asynchronous device-pointer search + a stream synchronise, for a small index (the exact scan: a few thousand frames is what
the caller's videos give) and the 1M-row one (fp16 streaming scan), integer and string ids."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ctypes import POINTER, c_int32
from video_quierer_amd import _lib
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex

_lib.init(0)
lib = _lib.load()
dev = torch.device("cuda", 0)


def p50(fn, reps=300):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e6 * ts[len(ts) // 2], 1e6 * ts[int(len(ts) * 0.95)]


print("VQ_AMD_HOST_FAST =", os.environ.get("VQ_AMD_HOST_FAST", "1 (default)"), flush=True)
for n in (4000, 100_000, 1_000_000):
    g = torch.Generator(device=dev); g.manual_seed(1)
    for ids_kind in ("int", "str"):
        idx = OptimizedHNSWIndex(dimension=512)
        for c0 in range(0, n, 250_000):
            c = min(250_000, n - c0)
            blk = torch.randn((c, 512), device=dev, generator=g)
            torch.cuda.synchronize()
            ids = range(c0, c0 + c) if ids_kind == "int" else [f"video{r // 1000}_{r % 1000}" for r in range(c0, c0 + c)]
            idx.add_device(blk.data_ptr(), c, ids, normalize=True)
            idx.synchronize()
        q = torch.randn((64, 512), device=dev, generator=g)
        q = q / q.norm(dim=1, keepdim=True)
        qh = q.cpu().numpy()
        idx.search(qh[0], 20)
        for k in (10, 20):
            a = p50(lambda: idx.search(qh[3], k))
            oi, od = np.empty((1, k), np.int32), np.empty((1, k), np.float32)
            q1 = np.ascontiguousarray(qh[3:4])
            b = p50(lambda: lib.vq_index_search(idx._h, _lib.fptr(q1), 1, k, 0, oi.ctypes.data_as(POINTER(c_int32)), _lib.fptr(od)))
            di = torch.empty((1, k), dtype=torch.int32, device=dev); dd = torch.empty((1, k), device=dev)
            def dev_call():
                idx.search_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr()); idx.synchronize()
            c_ = p50(dev_call)
            print(f"N={n:>8} ids={ids_kind} k={k}: python search p50 {a[0]:6.1f} us (p95 {a[1]:6.1f}) | C vq_index_search {b[0]:6.1f} (p95 {b[1]:6.1f}) | "
                  f"search_device+sync {c_[0]:6.1f} (p95 {c_[1]:6.1f})", flush=True)
        idx.close()

# where the small-index (exact scan) search spends its device time, per kernel class
print("exact-scan path, one query, device time per class (us):", flush=True)
for n in (4000, 8192, 16000):
    g = torch.Generator(device=dev); g.manual_seed(2)
    idx = OptimizedHNSWIndex(dimension=512)
    blk = torch.randn((n, 512), device=dev, generator=g); torch.cuda.synchronize()
    idx.add_device(blk.data_ptr(), n, range(n), normalize=True); idx.synchronize()
    q = torch.randn((4, 512), device=dev, generator=g); q = q / q.norm(dim=1, keepdim=True)
    for k in (10, 20):
        di = torch.empty((1, k), dtype=torch.int32, device=dev); dd = torch.empty((1, k), device=dev)
        for _ in range(3):
            idx.search_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr())
        idx.profile_begin()
        for _ in range(20):
            idx.search_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr())
        pr = idx.profile_end()
        print(f"  N={n} k={k}: " + ", ".join(f"{c} {1e3 * v['ms'] / 20:.1f}" for c, v in pr.items() if v["launches"]), flush=True)
    idx.close()

#!/usr/bin/env python3
"""Small-batch search latency over 1M x 512 (device-resident, asynchronous on the index's stream): wall time per search
over 200 back-to-back searches and the per-kernel-class device time, for nq in {1, 2, 4, 5, 16, 32} and k in {10, 20, 40}."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex
n, D = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
idx = OptimizedHNSWIndex(dimension=D)
for c0 in range(0, n, 250_000):
    c = min(250_000, n - c0)
    blk = torch.randn((c, D), device=dev, generator=g)
    torch.cuda.synchronize()
    idx.add_device(blk.data_ptr(), c, range(c0, c0 + c), normalize=True)
    idx.synchronize()
q = torch.randn((64, D), device=dev, generator=g); q = q / q.norm(dim=1, keepdim=True)
for k in (10, 20, 40):
    ids = torch.empty((64, k), dtype=torch.int32, device=dev); dd = torch.empty((64, k), device=dev)
    for nq in (1, 2, 4, 5, 16, 32):
        for _ in range(10):
            idx.search_device(q.data_ptr(), nq, k, ids.data_ptr(), dd.data_ptr())
        idx.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            idx.search_device(q.data_ptr(), nq, k, ids.data_ptr(), dd.data_ptr())
        idx.synchronize()
        wall = (time.perf_counter() - t0) / 200
        idx.profile_begin()
        for _ in range(20):
            idx.search_device(q.data_ptr(), nq, k, ids.data_ptr(), dd.data_ptr())
        prof = {kk: round(v["ms"] / 20 * 1e3, 1) for kk, v in idx.profile_end().items() if v["launches"]}
        print(f"N={n} k={k:2d} nq={nq:2d}: {wall*1e6:7.1f} us per search = {2.0*n*D/wall/1e12:5.2f} TB/s of the fp16 matrix ({2.0*n*D/wall/8e12:.3f} of 8 TB/s); "
              f"kernel classes (us, event brackets) {prof}; stats {idx.last_search_stats()}", flush=True)

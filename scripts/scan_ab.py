#!/usr/bin/env python3
"""A/B of the batch scan kernels on one box: per-kernel device time of a top-10 search over N x 512 at Q queries.
usage: scan_ab.py [N] [Q] [k] [dim]   (env VQ_AMD_SCAN = 2 four-phase | 4 deep prefetch)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex, MODE_FP16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
D = int(sys.argv[4]) if len(sys.argv) > 4 else 512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
idx = OptimizedHNSWIndex(dimension=D)
for c0 in range(0, n, 250_000):
    c = min(250_000, n - c0)
    blk = torch.randn((c, D), device=dev, generator=g)
    torch.cuda.synchronize()
    idx.add_device(blk.data_ptr(), c, range(c0, c0 + c), normalize=True)
    idx.synchronize()
q = torch.randn((nq, D), device=dev, generator=g); q = q / q.norm(dim=1, keepdim=True)
ids = torch.empty((nq, K), dtype=torch.int32, device=dev); dd = torch.empty((nq, K), device=dev)
torch.cuda.synchronize()
for _ in range(2):
    idx.search_device(q.data_ptr(), nq, K, ids.data_ptr(), dd.data_ptr(), mode=MODE_FP16)
idx.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    idx.search_device(q.data_ptr(), nq, K, ids.data_ptr(), dd.data_ptr(), mode=MODE_FP16)
idx.synchronize()
wall = (time.perf_counter() - t0) / 5
idx.profile_begin()
for _ in range(5):
    idx.search_device(q.data_ptr(), nq, K, ids.data_ptr(), dd.data_ptr(), mode=MODE_FP16)
prof = {k: round(v["ms"] / 5, 4) for k, v in idx.profile_end().items() if v["launches"]}
print(f"VQ_AMD_SCAN={os.environ.get('VQ_AMD_SCAN', 'default')} N={n} dim={D} Q={nq} k={K}: {wall*1e3:.3f} ms per batch = {nq/wall:.0f} q/s; kernels {prof}; "
      f"stats {idx.last_search_stats()}; checksum {int(ids.long().sum())} {float(dd.double().sum()):.9f}")

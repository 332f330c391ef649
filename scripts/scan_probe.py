#!/usr/bin/env python3
"""One fp16-scan search at N rows x Q queries (for rocprofv3).  usage: scan_probe.py [N] [Q] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex, MODE_FP16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
idx = OptimizedHNSWIndex(dimension=512)
for c0 in range(0, n, 250_000):
    c = min(250_000, n - c0)
    blk = torch.randn((c, 512), device=dev, generator=g)
    idx.add_device(blk.data_ptr(), c, range(c0, c0 + c), normalize=True)
    torch.cuda.synchronize()
q = torch.randn((nq, 512), device=dev, generator=g); q = q / q.norm(dim=1, keepdim=True)
ids = torch.empty((nq, 10), dtype=torch.int32, device=dev); dd = torch.empty((nq, 10), device=dev)
for _ in range(reps):
    idx.search_device(q.data_ptr(), nq, 10, ids.data_ptr(), dd.data_ptr(), mode=MODE_FP16)
    idx.synchronize()
print("stats", idx.last_search_stats())

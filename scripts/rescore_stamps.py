#!/usr/bin/env python3
"""Where the single-query re-score pass spends its cycles: s_memtime at the phase boundaries of rescore_verify_small_kernel
(query 0's workgroup).  Needs the diagnostic build: make -C video-quierer_amd/csrc STAMPS=1 OUT=../lib/libvq_amd_stamps.so
OBJDIR=../lib/obj_stamps, then VQ_AMD_LIB=$PWD/video-quierer_amd/lib/libvq_amd_stamps.so python scripts/rescore_stamps.py [N] [Q]."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex, MODE_FP16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
idx = OptimizedHNSWIndex(dimension=512)
for c0 in range(0, n, 250_000):
    c = min(250_000, n - c0)
    blk = torch.randn((c, 512), device=dev, generator=g)
    torch.cuda.synchronize()
    idx.add_device(blk.data_ptr(), c, range(c0, c0 + c), normalize=True)
    idx.synchronize()
q = torch.randn((nq, 512), device=dev, generator=g); q = q / q.norm(dim=1, keepdim=True)
ids = torch.empty((nq, 10), dtype=torch.int32, device=dev); dd = torch.empty((nq, 10), device=dev)
for _ in range(20):
    idx.search_device(q.data_ptr(), nq, 10, ids.data_ptr(), dd.data_ptr(), mode=MODE_FP16)
idx.synchronize()
lib = _lib.load()
lib.vq_debug_dump_rescore_stamps.restype = ctypes.c_int
lib.vq_debug_dump_rescore_stamps()

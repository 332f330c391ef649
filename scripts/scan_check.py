#!/usr/bin/env python3
"""fp16 scan vs the C oracle on a few shapes (debug helper).  usage: VQ_AMD_SCAN=2 scan_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import knn_oracle
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex, MODE_FP16
rng = np.random.default_rng(31)
allv = knn_oracle.normalize_rows(rng.standard_normal((24576, 512)).astype(np.float32))
allq = knn_oracle.normalize_rows(rng.standard_normal((300, 512)).astype(np.float32))
for n, nq, k in ((16384, 256, 10), (20480, 256, 10), (20001, 256, 10), (20001, 130, 10), (18432, 17, 10), (24576, 300, 10)):
    idx = OptimizedHNSWIndex(dimension=512)
    idx.add_device  # noqa
    from video_quierer_amd import _lib
    _lib.check(_lib.load().vq_index_add(idx._h, _lib.fptr(allv[:n]), n, 0))
    idx._ids = list(range(n)); idx._row_of = {i: i for i in range(n)}; idx.element_count = n; idx.entry_point = 0
    idx.search_mode = MODE_FP16
    ids, d = idx._raw_search(allq[:nq], k)
    oid, od = knn_oracle.topk(allv[:n], allq[:nq], k)
    bad = np.where((ids != oid).any(axis=1))[0]
    print(f"n={n} nq={nq}: mismatching queries {len(bad)} {bad[:10].tolist()} stats {idx.last_search_stats()}")
    if len(bad):
        q = bad[0]
        print("   got ", ids[q].tolist()); print("   want", oid[q].tolist())
        miss = [int(r) for r in oid[q] if r not in ids[q]]
        print("   missing rows", miss, "row%2048", [r % 2048 for r in miss], "tile", [(r % 2048)//256 for r in miss], "in-tile", [r % 256 for r in miss])
    idx.close()

#!/usr/bin/env python3
"""Randomised parity hunt for the search path: index size, dimension, k, batch size, duplicate rows, id kind (row numbers, shuffled
integers, the caller's strings) and search mode drawn at random; every result list is compared with the reference's own rule —
`sorted((distance, id) ...)[:k]` (hnsw.py:269 / :518) over the C oracle's exact distances — ids and distances bit for bit.
    python scripts/fuzz_search.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from video_quierer_amd import _lib
from video_quierer_amd.indexes.hnsw import MODE_AUTO, MODE_EXACT, MODE_FP16, OptimizedHNSWIndex
from oracle import knn_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
_lib.init(0)
t_end, cases, checked = time.time() + budget, 0, 0
while time.time() < t_end:
    dim = int(rng.choice([64, 128, 256, 512, 768]))
    n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 5000), rng.integers(5000, 20000), rng.integers(16000, 45000)]))
    k = int(rng.choice([1, rng.integers(2, 21), rng.integers(21, 41), rng.integers(41, 101), rng.integers(101, 300)]))
    nq = int(rng.choice([1, 1, rng.integers(2, 9), rng.integers(9, 97), rng.integers(97, 140)]))
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    dup = rng.random() < 0.6
    if dup and n > 4:                                   # duplicate frames: groups of equal rows, some large
        for _ in range(int(rng.integers(1, 6))):
            src = int(rng.integers(0, n))
            grp = rng.integers(0, n, int(rng.choice([2, 3, 8, 60, min(n, 400)])))
            rows[grp] = rows[src]
    kind = rng.choice(["rows", "ints", "strs"])
    if kind == "rows":
        ids = list(range(n))
    elif kind == "ints":
        ids = [int(v) for v in rng.permutation(n) * 3 + 7]
    else:
        per = max(1, n // 3)
        ids = [f"video{r // per}_{r % per}" for r in range(n)]
    modes = [MODE_AUTO, MODE_EXACT] + ([MODE_FP16] if dim % 64 == 0 and k <= 100 else [])
    mode = int(rng.choice(modes))
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    if dup:
        pick = rng.integers(0, n, nq)
        near = rng.random(nq) < 0.5
        qs[near] = rows[pick[near]] + np.float32(0.02) * qs[near]
    idx = OptimizedHNSWIndex(dimension=dim)
    half = n // 2
    idx.add_batch(rows[:half], ids[:half]) if half else None
    idx.add_batch(rows[half:], ids[half:])
    idx.search_mode = mode
    res = [idx.search(qs[0], k)] if nq == 1 else idx.search_batch(list(qs), k)
    stored = idx._export()
    uq = np.stack([q / np.linalg.norm(q) for q in qs]).astype(np.float32)
    kk = min(k, n)
    for j in range(nq):
        d = knn_oracle.distances(stored, uq[j])
        kth = np.partition(d, kk - 1)[kk - 1]
        cand = np.nonzero(d <= kth)[0]
        want = sorted((d[r], ids[r]) for r in cand)[:kk]
        got = [(r["distance"], r["id"]) for r in res[j]]
        if got != want:
            bad = next(i for i, (a, b) in enumerate(zip(got + [None] * kk, want)) if a != b)
            print(f"MISMATCH seed={seed} case={cases}: n={n} dim={dim} k={k} nq={nq} ids={kind} mode={mode} dup={dup} query {j} rank {bad}: got {got[bad] if bad < len(got) else None} want {want[bad]}; stats {idx.last_search_stats()}")
            sys.exit(1)
        checked += 1
    idx.close()
    cases += 1
    if cases % 25 == 0:
        print(f"{cases} cases, {checked} result lists identical to the (distance, id) order over the oracle's distances", flush=True)
print(f"done: {cases} cases, {checked} result lists, all identical (seed {seed})")

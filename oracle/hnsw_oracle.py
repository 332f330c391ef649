"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Pure-Python restatement of the reference's HNSW index
(reference src/indexes/hnsw.py) — the *approximate* graph index whose
``search`` the product replaces with an exact GPU scan.  It exists so that

* the graph semantics the product deliberately drops are written down and
  pinned (same ``random.seed`` ⇒ same levels, same graph, same result lists as
  the real reference: tests/golden/knn_cfg1.npz, captured by
  tests/golden/make_golden.py), and
* recall of the default-``ef_search`` reference against the product's exact
  answer can be reported on boxes where the reference does not exist.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.

State is array-shaped (row-indexed lists) rather than the reference's
id-keyed dicts; ids are mapped to rows on insert.  Arithmetic follows the
reference exactly: distance = ``1.0 - np.dot(a, b)`` on fp32 unit vectors
(hnsw.py:59-66), level = ``int(-ln(U(0,1)) * mL)`` from the global ``random``
module (:68-74).
"""
from __future__ import annotations

import heapq
import math
import random
from typing import Dict, Hashable, List, Sequence, Tuple

import numpy as np


class HnswOracle:
    def __init__(self, dimension: int = 512, M: int = 16, ef_construction: int = 200,
                 ef_search: int = 50, max_M: int = 16,
                 level_generation_factor: float = 1.0 / math.log(2.0)):
        # hnsw.py:25-57
        self.dimension, self.M, self.max_M = dimension, M, max_M
        self.ef_construction, self.ef_search = ef_construction, ef_search
        self.mL = level_generation_factor
        self.ids: List[Hashable] = []          # row -> caller id
        self.row_of: Dict[Hashable, int] = {}  # caller id -> row
        self.vec: List[np.ndarray] = []        # row -> unit fp32 vector
        self.level: List[int] = []             # row -> top level
        self.adj: List[List[set]] = []         # row -> per-level neighbour rows
        self.entry = -1

    # -- primitives ---------------------------------------------------------
    def _d(self, a: np.ndarray, b: np.ndarray):
        return 1.0 - np.dot(a, b)                                   # hnsw.py:66

    def _draw_level(self) -> int:
        return int(-math.log(random.uniform(0, 1)) * self.mL)       # hnsw.py:72-74

    def _key(self, row: int):
        # the reference's heaps hold (distance, caller_id): ties on distance are
        # broken by comparing caller ids, so heap entries carry the id, not the row
        return self.ids[row]

    def _beam(self, q: np.ndarray, starts: Sequence[int], width: int, lv: int) -> List[Tuple[float, int]]:
        """Best-first layer search (hnsw.py:76-121).  Returns ≤width (dist,row), unsorted."""
        seen = set()
        frontier: list = []   # min-heap (dist, id, row)
        best: list = []       # max-heap (-dist, id, row), capped at width
        for r in starts:
            d = self._d(q, self.vec[r])
            heapq.heappush(frontier, (d, self._key(r), r))
            heapq.heappush(best, (-d, self._key(r), r))
            seen.add(r)
        while frontier:
            d, _, r = heapq.heappop(frontier)
            if best and d > -best[0][0]:                            # :103-104
                break
            nbrs = self.adj[r][lv] if lv < len(self.adj[r]) else ()
            for nb in nbrs:
                if nb in seen:
                    continue
                seen.add(nb)
                dn = self._d(q, self.vec[nb])
                if len(best) < width or dn < -best[0][0]:           # :113
                    heapq.heappush(frontier, (dn, self._key(nb), nb))
                    heapq.heappush(best, (-dn, self._key(nb), nb))
                    if len(best) > width:
                        heapq.heappop(best)
        return [(-nd, r) for nd, _, r in best]

    def _closest(self, cands: List[Tuple[float, int]], m: int) -> List[int]:
        """'Heuristic' selection = the m closest (hnsw.py:123-148)."""
        if len(cands) <= m:
            return [r for _, r in cands]
        ordered = sorted(cands, key=lambda t: (t[0], self._key(t[1])))
        return [r for _, r in ordered[:m]]

    # -- build ----------------------------------------------------------------
    def add(self, vector: np.ndarray, node_id: Hashable) -> None:
        """hnsw.py:150-229."""
        v = vector / np.linalg.norm(vector)                         # :157 (no zero guard)
        lvl = self._draw_level()
        if node_id in self.row_of:        # re-adding an id overwrites its slot (dict semantics, :160-166)
            row = self.row_of[node_id]
            self.vec[row], self.level[row] = v, lvl
            old = self.adj[row]
            self.adj[row] = [set() for _ in range(lvl + 1)] + old[lvl + 1:]
        else:
            row = len(self.ids)
            self.ids.append(node_id)
            self.row_of[node_id] = row
            self.vec.append(v)
            self.level.append(lvl)
            self.adj.append([set() for _ in range(lvl + 1)])
        if self.entry < 0:                                          # :168-171
            self.entry = row
            return
        top = self.level[self.entry]
        near = [self.entry]
        for lv in range(max(top, lvl), lvl, -1):                    # :174-180 greedy descent
            near = [r for _, r in self._beam(v, near, 1, lv)]
        for lv in range(min(lvl, top), -1, -1):                     # :183-223
            cands = self._beam(v, near, self.ef_construction, lv)
            near = [r for _, r in cands]
            cap = self.M if lv > 0 else self.max_M
            for nb in self._closest(cands, cap):
                self.adj[row][lv].add(nb)
                self.adj[nb][lv].add(row)
                links = list(self.adj[nb][lv])
                if len(links) > cap:                                # prune the over-full neighbour
                    scored = [(self._d(self.vec[nb], self.vec[c]), c) for c in links]
                    keep = set(self._closest(scored, cap))
                    self.adj[nb][lv] = keep
                    for c in links:
                        if c not in keep:
                            self.adj[c][lv].discard(nb)
        if lvl > top:                                               # :226-227
            self.entry = row

    def add_batch(self, vectors, node_ids) -> None:
        for v, i in zip(vectors, node_ids):                         # :231-236
            self.add(v, i)

    # -- query ----------------------------------------------------------------
    def search(self, query: np.ndarray, k: int = 5) -> List[dict]:
        """hnsw.py:488-528 (≡ :238-280: the 'optimized' layer search always falls
        back to the plain one because every call passes one entry point, :445-446)."""
        if self.entry < 0:
            return []
        q = query / np.linalg.norm(query)
        near = [self.entry]
        for lv in range(self.level[self.entry], 0, -1):
            near = [r for _, r in self._beam(q, near, 1, lv)]
        cands = self._beam(q, near, max(self.ef_search, k), 0)
        ranked = sorted(((d, self.ids[r]) for d, r in cands))[:k]   # (distance, id) ascending, :518
        return [{"id": i, "distance": d, "score": 1.0 - d} for d, i in ranked]

    def size(self) -> int:
        return len(self.ids)


def brute_force(rows: np.ndarray, ids: Sequence[Hashable], query: np.ndarray, k: int) -> List[dict]:
    """What the reference returns once ``ef_search >= N`` (SURVEY.md §8a "same result"):
    k smallest of fp32(1 - fp32(dot(X_i, q))) ordered by (distance, id).
    ``rows`` must already be the stored (normalised) vectors."""
    q = query / np.linalg.norm(query)
    d = np.float32(1.0) - rows.astype(np.float32) @ q.astype(np.float32)
    order = sorted(range(len(ids)), key=lambda r: (d[r], ids[r]))[:k]
    return [{"id": ids[r], "distance": d[r], "score": np.float32(1.0) - d[r]} for r in order]

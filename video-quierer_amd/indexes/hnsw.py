"""Drop-in for the reference's ``indexes.hnsw`` module (reference
src/indexes/hnsw.py): ``HNSWIndex`` / ``OptimizedHNSWIndex`` with the same
constructor, methods, result dicts and pickle layout — but ``search`` is an
EXACT scan of a device-resident matrix in libvq_amd (no graph walk), i.e. what
the reference itself returns once ``ef_search >= N``:

    distance_i = fp32(1 - fp32(dot(x_i, q))),  k smallest by (distance, id).

The graph parameters (M, ef_construction, ef_search, max_M,
level_generation_factor) are accepted and echoed by ``get_stats`` but have no
effect.  There is no CPU path: without the library or a gfx950 device
construction raises.
"""
from __future__ import annotations

import ctypes
import hashlib
import math
import os
import pickle
import threading
import time
from collections.abc import Mapping
from concurrent.futures import ThreadPoolExecutor
from ctypes import POINTER, c_float, c_int32, c_int64, c_void_p
from typing import Dict, Hashable, List, Optional, Sequence

import numpy as np

from video_quierer_amd import _lib

MODE_AUTO, MODE_EXACT, MODE_FP16 = 0, 1, 2


class _RowView(Mapping):
    """Read-only ``id -> stored row`` view of an index (``HNSWIndex.data``)."""

    def __init__(self, index):
        self._index = index

    def __len__(self):
        return len(self._index._ids)

    def __iter__(self):
        return iter(self._index._ids)

    def __contains__(self, node_id):
        return node_id in self._index._row_of

    def __getitem__(self, node_id):
        idx = self._index
        with idx.lock:
            row = idx._row_of[node_id]                       # KeyError for an unknown id, as a dict would
            out = np.empty((1, idx.dimension), dtype=np.float32)
            rn = (ctypes.c_int64 * 1)(row)
            _lib.check(_lib.load().vq_index_read_rows(idx._h, rn, 1, _lib.fptr(out)))
        return out[0]

    def items(self):
        rows = self._index._export()
        return [(i, rows[r]) for r, i in enumerate(self._index._ids)]

    def values(self):
        rows = self._index._export()
        return [rows[r] for r in range(len(self._index._ids))]


class HNSWIndex:
    def __init__(self, dimension: int = 512, M: int = 16, ef_construction: int = 200, ef_search: int = 50,
                 max_M: int = 16, level_generation_factor: float = 1.0 / math.log(2.0), num_threads: int = 4,
                 device: Optional[int] = None):
        # reference :25-57
        self.dimension = dimension
        self.M = M
        self.max_M = max_M
        self.ef_construction = ef_construction
        self.ef_search = ef_search
        self.level_generation_factor = level_generation_factor
        self.num_threads = num_threads

        self.entry_point = None
        self.element_count = 0
        self.lock = threading.RLock()
        self.thread_pool = ThreadPoolExecutor(max_workers=num_threads)
        self.build_time = 0
        self.search_times: List[float] = []
        self.search_mode = MODE_AUTO

        self._ids: List[Hashable] = []            # row -> caller id
        self._row_of: Dict[Hashable, int] = {}    # caller id -> row
        self._identity = True                     # ids are exactly 0..n-1 in row order
        # (distance, id) tie order on the device (vq_index_set_id_ranks): "stale" until the ranks of the current ids
        # have been uploaded, "device" afterwards, "host" when the ids cannot be put in one order (see _sync_tie_order)
        self._tie_order = "device"
        _lib.init(device)
        h = c_void_p()
        _lib.check(_lib.load().vq_index_create(int(dimension), ctypes.byref(h)))
        self._h = h

    # -- reference-shaped views of the state (hnsw.py:44-49) ---------------------
    @property
    def data(self) -> "Mapping[Hashable, np.ndarray]":
        """The reference's ``self.data`` dict (id -> stored unit vector, hnsw.py:44) as a read-only view of the device matrix:
        a lookup fetches that one row, iteration / ``len`` / ``in`` touch no device memory, ``items()`` / ``values()`` / ``dict(...)``
        export the matrix once.  (Round 2 exported every row on every access.)"""
        return _RowView(self)

    @property
    def levels(self) -> Dict[Hashable, int]:
        return {i: 0 for i in self._ids}          # exact index: a single flat level

    @property
    def graph(self) -> dict:
        return {0: {i: set() for i in self._ids}}

    def _export(self) -> np.ndarray:
        n = len(self._ids)
        rows = np.empty((n, self.dimension), dtype=np.float32)
        if n:
            _lib.check(_lib.load().vq_index_export(self._h, _lib.fptr(rows)))
        return rows

    # -- build --------------------------------------------------------------------
    @staticmethod
    def _unit(vector) -> np.ndarray:
        v = np.asarray(vector)
        return (v / np.linalg.norm(v)).astype(np.float32, copy=False)     # hnsw.py:157 (no zero guard)

    @classmethod
    def _unit_rows(cls, vectors) -> np.ndarray:
        """Row-wise ``v / np.linalg.norm(v)`` for a block (reference :157, :250) with the per-row Python loop taken out.
        For a 1-D float32 vector ``np.linalg.norm`` is ``sqrt(v.dot(v))`` — BLAS sdot; ``np.matmul`` of a stack of
        (1, d) @ (d, 1) products runs that same dot per row, so the norms (and ``v / norm`` in float32) are the reference's
        bits, 6x faster than the loop.  That equivalence is numpy's implementation, not its contract: a sample of rows (a
        fixed spread + 16 random ones per call) is checked against ``np.linalg.norm`` on every call and any difference (or
        any other dtype) takes the per-row loop.  A SAMPLE: "stored rows == reference .data" is asserted on whole matrices
        by the golden tests (1k rows bit for bit, sha256 of all rows at 10k and 100k), not by this guard."""
        vs = vectors if isinstance(vectors, np.ndarray) else None
        if vs is None:
            try:
                vs = np.asarray(vectors)
            except ValueError:
                vs = None
        if vs is None or vs.ndim != 2 or vs.dtype != np.float32 or vs.shape[0] < 8:
            return np.stack([cls._unit(v) for v in vectors])
        vs = np.ascontiguousarray(vs)
        with np.errstate(invalid="ignore", divide="ignore"):
            norms = np.sqrt(np.matmul(vs[:, None, :], vs[:, :, None]).reshape(-1))
            n = vs.shape[0]
            # a fixed spread of rows plus a fresh random set on every call: the equivalence is sampled, not proven — a row
            # outside both sets that differed would go unnoticed (numpy would have to pick another kernel for some rows of ONE
            # matmul: its batched (1, d) @ (d, 1) loop calls the same dot per row), which is why every golden test also
            # compares the stored rows with the reference's `.data` bit for bit, whole matrices, sha256 at 10k / 100k rows
            probe = np.unique(np.concatenate([[0, n - 1], np.linspace(0, n - 1, 16).astype(np.int64),
                                              np.random.default_rng().integers(0, n, 16)]))
            for i in probe:
                ref = np.linalg.norm(vs[i])
                if not (norms[i] == ref or (np.isnan(norms[i]) and np.isnan(ref))):
                    return np.stack([cls._unit(v) for v in vs])
            return vs / norms[:, None]

    @staticmethod
    def _ids_are_rows(ids: Sequence[Hashable], base: int) -> bool:
        """Are `ids` exactly base, base + 1, ... (integers)?  Ids are any hashable: a ragged mix (tuples of different lengths, a
        str beside a tuple) makes ``np.asarray`` raise — that is simply "no"."""
        if not ids or not isinstance(ids[0], (int, np.integer)) or not isinstance(ids[-1], (int, np.integer)):
            return False
        if isinstance(ids, range):
            return ids.step == 1 and ids.start == base
        try:
            arr = np.asarray(ids)
        except (ValueError, TypeError):
            return False
        return bool(arr.ndim == 1 and arr.dtype.kind in "iu" and np.array_equal(arr, np.arange(base, base + len(ids))))

    def _sync_tie_order(self) -> None:
        """The reference returns ``sorted(candidates)[:k]`` over (distance, id) tuples (hnsw.py:269 / :518): rows at equal
        distance come back in id order.  The device orders by (distance, rank of the row's id) once it has the ranks
        (vq_index_set_id_ranks); they are recomputed here, lazily, by the first search after new ids came in.  ``sorted``
        compares ids only where distances tie, so the reference tolerates ids that have no common order (an int beside a
        str) as long as no tie meets them; a global ranking cannot — such an index keeps the host-side path (over-fetch,
        re-sort per query)."""
        if self._tie_order != "stale":
            return
        ids = self._ids
        n = len(ids)
        order = None
        if type(ids[0]) is str and type(ids[-1]) is str and all(type(i) is str for i in ids):
            try:
                arr = np.array(ids)                                  # '<U..': numpy compares code points, as str does
                if arr.dtype.kind == "U" and arr.shape == (n,):
                    cand = np.argsort(arr, kind="stable")
                    srt = arr[cand]
                    if n < 2 or bool(np.all(srt[1:] > srt[:-1])):    # strict: two ids numpy cannot tell apart (trailing NULs) -> Python
                        order = cand
            except (ValueError, TypeError):
                order = None
        if order is None:
            try:
                order = np.fromiter(sorted(range(n), key=ids.__getitem__), dtype=np.int64, count=n)
            except TypeError:                                        # no total order over these ids
                _lib.check(_lib.load().vq_index_set_id_ranks(self._h, None, 0))
                self._tie_order = "host"
                return
        rank = np.empty(n, dtype=np.int32)
        rank[order] = np.arange(n, dtype=np.int32)
        _lib.check(_lib.load().vq_index_set_id_ranks(self._h, rank.ctypes.data_as(POINTER(c_int32)), n))
        self._tie_order = "device"

    def add(self, vector: np.ndarray, node_id: Hashable) -> None:
        self.add_batch([vector], [node_id])                               # reference :150-229

    def add_batch(self, vectors: Sequence[np.ndarray], node_ids: Sequence[Hashable]) -> None:
        """Reference :231-236 (a per-vector loop there); here one device append."""
        node_ids = list(node_ids)
        n = min(len(vectors), len(node_ids))                              # zip semantics
        if n == 0:
            return
        t0 = time.time()
        with self.lock:
            vs = np.asarray(vectors[:n] if not isinstance(vectors, np.ndarray) else vectors[:n])
            if vs.ndim != 2 or vs.shape[1] != self.dimension:
                raise ValueError(f"vectors must be [n,{self.dimension}], got {vs.shape}")
            # row-wise `v / np.linalg.norm(v)` exactly as the reference computes it (:157)
            unit = self._unit_rows(vs)
            unit = np.ascontiguousarray(unit, dtype=np.float32)
            ids_n = node_ids[:n]
            if self._row_of.keys().isdisjoint(ids_n) and len(set(ids_n)) == n:
                # the ingest case (video_search_system.py:168-176: every frame id is new): no per-id walk
                base = len(self._ids)
                identity = self._identity and self._ids_are_rows(ids_n, base)     # decided before anything is committed
                _lib.check(_lib.load().vq_index_add(self._h, _lib.fptr(unit), n, 0))
                self._row_of.update(zip(ids_n, range(base, base + n)))
                self._identity = identity
                if not identity:
                    self._tie_order = "stale"
                self._ids.extend(ids_n)
                self.element_count += n                                   # reference counts every add (:229)
                if self.entry_point is None:
                    self.entry_point = self._ids[0]
                self.build_time += time.time() - t0
                return
            fresh_rows, fresh_ids = [], []
            upd_rows, upd_src = [], []                                   # re-added ids: (stored row, batch position), in call order
            seen_now: Dict[Hashable, int] = {}
            for j, nid in enumerate(ids_n):
                if nid in self._row_of:
                    upd_rows.append(self._row_of[nid]); upd_src.append(j)
                elif nid in seen_now:                                    # later duplicate wins (dict semantics)
                    fresh_rows[seen_now[nid]] = j
                else:
                    seen_now[nid] = len(fresh_rows)
                    fresh_rows.append(j)
                    fresh_ids.append(nid)
                self.element_count += 1                                   # reference counts every add (:229)
            if upd_rows:
                self._overwrite(upd_rows, unit if len(upd_src) == n else np.ascontiguousarray(unit[upd_src]))
            if fresh_rows:
                block = unit if len(fresh_rows) == n else np.ascontiguousarray(unit[fresh_rows])
                _lib.check(_lib.load().vq_index_add(self._h, _lib.fptr(block), len(fresh_rows), 0))
                base = len(self._ids)
                for j, nid in enumerate(fresh_ids):
                    self._row_of[nid] = base + j
                    if self._identity and not (isinstance(nid, (int, np.integer)) and int(nid) == base + j):
                        self._identity = False
                self._ids.extend(fresh_ids)
                if not self._identity:
                    self._tie_order = "stale"
            if self.entry_point is None and self._ids:
                self.entry_point = self._ids[0]
        self.build_time += time.time() - t0

    def add_device(self, d_rows: int, n: int, node_ids: Sequence[Hashable], normalize: bool = True) -> None:
        """Append n device-resident fp32 rows (e.g. straight from the encoder) without a host round trip."""
        node_ids = list(node_ids)
        if len(node_ids) != n:
            raise ValueError("node_ids must have n entries")
        with self.lock:
            if not self._row_of.keys().isdisjoint(node_ids) or len(set(node_ids)) != n:
                raise ValueError("add_device: ids must be new and unique")
            base = len(self._ids)
            identity = self._identity and self._ids_are_rows(node_ids, base)
            _lib.check(_lib.load().vq_index_add_device(self._h, c_void_p(d_rows), n, int(bool(normalize))))
            self._row_of.update(zip(node_ids, range(base, base + n)))       # (a per-id loop here cost 45 ms per 250k rows)
            self._identity = identity
            if not identity:
                self._tie_order = "stale"
            self._ids.extend(node_ids)
            self.element_count += n
            if self.entry_point is None and self._ids:
                self.entry_point = self._ids[0]

    def _overwrite(self, rows: Sequence[int], unit_vecs: np.ndarray) -> None:
        """Re-adding an id replaces its vector (a dict assignment in the reference, hnsw.py:160): the stored rows are
        replaced in place on the device — fp32 master, fp16 scan copy, norm range — in one call; a row named twice
        keeps its last vector, as sequential assignments would leave it."""
        rn = np.ascontiguousarray(rows, dtype=np.int64)
        _lib.check(_lib.load().vq_index_update_rows(self._h, _lib.fptr(unit_vecs), rn.ctypes.data_as(POINTER(ctypes.c_int64)),
                                                    len(rn), 0))

    # -- query --------------------------------------------------------------------
    def _raw_search(self, unit_queries: np.ndarray, k: int):
        nq = unit_queries.shape[0]
        ids = np.empty((nq, k), dtype=np.int32)
        dist = np.empty((nq, k), dtype=np.float32)
        _lib.check(_lib.load().vq_index_search(self._h, _lib.fptr(unit_queries), nq, k, int(self.search_mode),
                                               ids.ctypes.data_as(POINTER(c_int32)), _lib.fptr(dist)))
        return ids, dist

    def _search_many(self, queries: Sequence[np.ndarray], k: int) -> List[List[Dict]]:
        n = len(self._ids)
        unit = np.ascontiguousarray(self._unit_rows(queries), dtype=np.float32)
        if unit.shape[1] != self.dimension:
            raise ValueError(f"query dimension {unit.shape[1]} != index dimension {self.dimension}")
        kk = min(k, n)
        if kk <= 0:
            return [[] for _ in queries]          # reference: sorted(candidates)[:0] == [] (hnsw.py:269)
        if self._identity:
            ids, dist = self._raw_search(unit, kk)
            return [[{"id": int(i), "distance": d, "score": np.float32(1.0) - d} for i, d in zip(ri, rd) if i >= 0]
                    for ri, rd in zip(ids, dist)]
        self._sync_tie_order()
        if self._tie_order == "device":
            # the caller's ids (strings f"{video_id}_{i}", video_search_system.py:164-166): the device already ordered by
            # (distance, id rank) — exactly k results fetched, rows mapped to ids, nothing re-sorted
            ids, dist = self._raw_search(unit, kk)
            names = self._ids
            return [[{"id": names[i], "distance": d, "score": np.float32(1.0) - d} for i, d in zip(ri.tolist(), rd) if i >= 0]
                    for ri, rd in zip(ids, dist)]
        # ids without a common order: the library orders ties by row; the reference orders them by id
        # (hnsw.py:269/518).  Over-fetch until no tie group is cut at rank k, then re-sort.
        fetch = min(n, kk + 8)
        while True:
            ids, dist = self._raw_search(unit, fetch)
            cut = fetch < n and np.any(dist[:, kk - 1] == dist[:, fetch - 1])
            if not cut:
                break
            fetch = min(n, fetch * 2)
        out = []
        for ri, rd in zip(ids, dist):
            cand = sorted(((d, self._ids[i]) for i, d in zip(ri, rd) if i >= 0))[:kk]
            out.append([{"id": i, "distance": d, "score": np.float32(1.0) - d} for d, i in cand])
        return out

    def search(self, query: np.ndarray, k: int = 5) -> List[Dict]:
        """Reference :238-280 / :488-528."""
        if self.entry_point is None or self.element_count == 0:
            return []
        t0 = time.time()
        with self.lock:
            res = self._search_many([query], k)[0]
        self.search_times.append((time.time() - t0) * 1000)
        return res

    def search_batch(self, queries: List[np.ndarray], k: int = 5) -> List[List[Dict]]:
        """Reference :282-300 (a thread fan-out serialised by the index lock); here one batched scan."""
        if len(queries) == 0:
            return []
        if self.entry_point is None or self.element_count == 0:
            return [[] for _ in queries]
        t0 = time.time()
        with self.lock:
            res = self._search_many(queries, k)
        per = (time.time() - t0) * 1000 / len(queries)
        self.search_times.extend([per] * len(queries))
        return res

    def search_device(self, d_queries: int, nq: int, k: int, d_ids: int, d_dist: int, mode: Optional[int] = None) -> None:
        """Device pointers in/out (unit fp32 queries; int32 ROW numbers; fp32 distances); asynchronous.  Rows at equal
        distance come back in the order of their ids (as `search` returns them) unless the ids have no common order."""
        with self.lock:
            if not self._identity:
                self._sync_tie_order()
            _lib.check(_lib.load().vq_index_search_device(self._h, c_void_p(d_queries), int(nq), int(k),
                                                          int(self.search_mode if mode is None else mode),
                                                          c_void_p(d_ids), c_void_p(d_dist)))

    def synchronize(self) -> None:
        _lib.check(_lib.load().vq_index_synchronize(self._h))

    def set_stream(self, hip_stream: int) -> None:
        _lib.check(_lib.load().vq_index_set_stream(self._h, c_void_p(hip_stream or None)))

    def size(self) -> int:
        return self.element_count                                          # reference :302-304

    # -- persistence (reference :306-380: pickle + sha256 sidecar) ------------------
    def save(self, filepath: str) -> None:
        """Reference :306-339: same pickle keys, HIGHEST_PROTOCOL, SHA-256 sidecar.  The index is exact, so ``levels``
        and ``graph`` are flat: the reference class LOADS such a file (same rows, ids and parameters) but its graph
        walk has nothing to walk — a reference user re-adds the rows; files written by the
        reference load here and answer as the reference does.  The extra key ``exact_index`` marks files from here."""
        with self.lock:
            save_data = {
                "dimension": self.dimension, "M": self.M, "max_M": self.max_M,
                "ef_construction": self.ef_construction, "ef_search": self.ef_search,
                "level_generation_factor": self.level_generation_factor,
                "data": dict(self.data.items()), "levels": self.levels, "graph": self.graph,
                "entry_point": self.entry_point, "element_count": self.element_count,
                "exact_index": True,     # extra key: no navigable graph in this file
            }
            d = os.path.dirname(filepath)
            if d:
                os.makedirs(d, exist_ok=True)
            with open(filepath, "wb") as f:
                pickle.dump(save_data, f, protocol=pickle.HIGHEST_PROTOCOL)
            with open(filepath, "rb") as f:
                checksum = hashlib.sha256(f.read()).hexdigest()
            with open(filepath + ".sha256", "w") as f:
                f.write(checksum)

    def load(self, filepath: str) -> None:
        """Reference :341-380 (checksum check, missing sidecar tolerated, ValueError on mismatch).  Only ``data`` (and
        the parameters) are used; the rows are stored as given and MEASURED by the library (vq_index_add
        normalize=0): a file whose rows are not unit-norm still searches exactly, on the fp32 scan."""
        try:
            with open(filepath, "rb") as f:
                current = hashlib.sha256(f.read()).hexdigest()
            with open(filepath + ".sha256", "r") as f:
                expected = f.read().strip()
            if current != expected:
                raise ValueError("Index file corrupted (checksum mismatch)")
        except FileNotFoundError:
            print("Warning: No checksum file found, skipping verification")
        with open(filepath, "rb") as f:
            s = pickle.load(f)
        with self.lock:
            if s["dimension"] != self.dimension:
                lib = _lib.load()
                lib.vq_index_destroy(self._h)
                h = c_void_p()
                _lib.check(lib.vq_index_create(int(s["dimension"]), ctypes.byref(h)))
                self._h = h
            else:
                _lib.check(_lib.load().vq_index_clear(self._h))
            self.dimension = s["dimension"]
            self.M, self.max_M = s["M"], s["max_M"]
            self.ef_construction, self.ef_search = s["ef_construction"], s["ef_search"]
            self.level_generation_factor = s["level_generation_factor"]
            self._ids, self._row_of, self._identity, self._tie_order = [], {}, True, "device"
            ids = list(s["data"].keys())
            if ids:
                rows = np.ascontiguousarray(np.stack([np.asarray(s["data"][i], dtype=np.float32) for i in ids]))
                _lib.check(_lib.load().vq_index_add(self._h, _lib.fptr(rows), len(ids), 0))   # stored rows are already unit
                for r, nid in enumerate(ids):
                    self._row_of[nid] = r
                    if self._identity and not (isinstance(nid, (int, np.integer)) and int(nid) == r):
                        self._identity = False
                self._ids = ids
                if not self._identity:
                    self._tie_order = "stale"
            self.entry_point = s["entry_point"]
            self.element_count = s["element_count"]

    # -- stats (reference :382-402) ---------------------------------------------------
    def get_stats(self) -> Dict:
        if self.search_times:
            avg = sum(self.search_times) / len(self.search_times)
            p95 = np.percentile(self.search_times, 95)
        else:
            avg = p95 = 0
        return {
            "element_count": self.element_count,
            "entry_point_level": 0,
            "avg_search_time_ms": avg,
            "p95_search_time_ms": p95,
            "total_searches": len(self.search_times),
            "dimension": self.dimension,
            "M": self.M,
            "ef_search": self.ef_search,
        }

    def last_search_stats(self) -> Dict[str, int]:
        st = (c_int64 * 3)()
        _lib.check(_lib.load().vq_index_last_search_stats(self._h, st))
        return {"verified": int(st[0]), "rescanned": int(st[1]), "exact_fallback": int(st[2])}

    def profile_begin(self) -> None:
        _lib.check(_lib.load().vq_index_profile_begin(self._h))

    def profile_end(self) -> Dict[str, dict]:
        lib = _lib.load()
        ms = (c_float * _lib.IDX_NCLASS)()
        cnt = (ctypes.c_int * _lib.IDX_NCLASS)()
        _lib.check(lib.vq_index_profile_end(self._h, ms, cnt))
        return {lib.vq_index_profile_class_name(i).decode(): {"ms": float(ms[i]), "launches": int(cnt[i])}
                for i in range(_lib.IDX_NCLASS)}

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().vq_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OptimizedHNSWIndex(HNSWIndex):
    """Reference :405-528.  ``use_numpy_optimization`` is accepted for signature
    compatibility; both classes run the same device scan."""

    def __init__(self, *args, use_numpy_optimization=True, **kwargs):
        super().__init__(*args, **kwargs)
        self.use_numpy_optimization = use_numpy_optimization

"""Drop-in for the live system's brute-force index, ``SimpleVideoIndex``
(reference video_search_overhaul.py:23-106) — SURVEY.md §8f "next" #2 — on the
same device scan as indexes.hnsw:

* ``add_frame(embedding, video_name, timestamp)`` appends the embedding AS GIVEN
  (the reference does not normalise stored rows, :31-38);
* ``search(query, k)`` = top-k of ``E @ (q / (||q|| + 1e-10))`` in descending
  similarity, ``np.argsort(sim)[::-1][:k]`` (:40-64), i.e. ties resolve to the
  LARGER frame id; results are the metadata dicts plus ``'score'`` (float);
* ``save_to_disk`` / ``load_from_disk`` keep the reference's pickle layout
  ``{embeddings, metadata, video_hashes, version}`` (:66-106).

Scores come back as ``1 - (1 - dot)`` from the distance the scan returns, so
they equal the reference's fp32 dot to ~1e-7 (not bit-for-bit).  Rows are pushed
to the GPU lazily, at the first search after an add.
"""
from __future__ import annotations

import logging
import pickle
from pathlib import Path
from typing import Dict, List

import numpy as np

from video_quierer_amd.indexes.hnsw import MODE_AUTO, HNSWIndex

logger = logging.getLogger(__name__)


class SimpleVideoIndex:
    def __init__(self):
        self.embeddings: List[np.ndarray] = []
        self.metadata: List[Dict] = []
        self.video_hashes: Dict = {}
        self._dev = None          # HNSWIndex used as the raw device matrix
        self._pushed = 0

    def add_frame(self, embedding: np.ndarray, video_name: str, timestamp: float):
        self.embeddings.append(embedding.astype(np.float32))
        self.metadata.append({"video_name": video_name, "timestamp": timestamp,
                              "frame_id": len(self.embeddings) - 1})

    def _sync_device(self) -> None:
        n = len(self.embeddings)
        if self._dev is not None and self._pushed > n:      # list was replaced/shrunk: rebuild
            self._dev.close()
            self._dev, self._pushed = None, 0
        if n == self._pushed:
            return
        block = np.ascontiguousarray(np.vstack(self.embeddings[self._pushed:]), dtype=np.float32)
        if self._dev is None:
            self._dev = HNSWIndex(dimension=block.shape[1])
        from video_quierer_amd import _lib
        _lib.check(_lib.load().vq_index_add(self._dev._h, _lib.fptr(block), block.shape[0], 0))   # stored as given
        # ids = -frame_id: the scan's (distance, id) tie rule then yields the larger frame first,
        # like the reference's reversed argsort
        self._dev._ids.extend(-i for i in range(self._pushed, n))
        self._dev._identity = False
        self._dev.element_count = n
        self._dev.entry_point = 0
        self._pushed = n

    def search(self, query_embedding: np.ndarray, k: int = 5) -> List[Dict]:
        if not self.embeddings:
            return []
        self._sync_device()
        q = np.asarray(query_embedding)
        query_norm = (q / (np.linalg.norm(q) + 1e-10)).astype(np.float32)          # reference :50-51
        dev = self._dev
        # rows are stored as given: the library measures |row|^2 of what it holds and only takes its fp16 scan
        # while the matrix is near-unit (include/vq_amd.h vq_index_add), so un-normalised embeddings stay exact
        dev.search_mode = MODE_AUTO
        n, kk = len(self.embeddings), min(k, len(self.embeddings))
        unit = np.ascontiguousarray(query_norm[None, :])
        fetch = min(n, kk + 8)
        while True:                                   # do not cut a tie group at rank k
            rows, dist = dev._raw_search(unit, fetch)
            if fetch >= n or dist[0, kk - 1] != dist[0, fetch - 1]:
                break
            fetch = min(n, fetch * 2)
        order = sorted(((d, -int(r)) for r, d in zip(rows[0], dist[0]) if r >= 0))[:kk]
        results = []
        for d, neg in order:
            md = self.metadata[-neg].copy()
            md["score"] = float(np.float32(1.0) - d)
            results.append(md)
        return results

    def save_to_disk(self, cache_path: Path):
        try:
            with open(cache_path, "wb") as f:
                pickle.dump({"embeddings": self.embeddings, "metadata": self.metadata,
                             "video_hashes": self.video_hashes, "version": "1.0"}, f)
            logger.info(f"Saved {len(self.embeddings)} embeddings to {cache_path}")
            return True
        except Exception as e:
            logger.error(f"Failed to save cache: {e}")
            return False

    def load_from_disk(self, cache_path: Path) -> bool:
        try:
            cache_path = Path(cache_path)
            if not cache_path.exists():
                return False
            with open(cache_path, "rb") as f:
                data = pickle.load(f)
            self.embeddings = data.get("embeddings", [])
            self.metadata = data.get("metadata", [])
            self.video_hashes = data.get("video_hashes", {})
            if self._dev is not None:
                self._dev.close()
            self._dev, self._pushed = None, 0
            logger.info(f"Loaded {len(self.embeddings)} embeddings from {cache_path}")
            return True
        except Exception as e:
            logger.error(f"Failed to load cache: {e}")
            return False

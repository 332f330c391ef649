"""Owner of a ``vq_comm`` handle (include/vq_amd.h): the path's two exchange steps over RCCL/xGMI, one process per
GPU — the all-gather of per-shard embeddings in front of ``index.add_batch`` (reference ingest loop
src/video_search_system.py:152-181) and the all-gather + (distance, id) merge of per-shard top-k for a row-sharded
matrix (reference src/indexes/hnsw.py:269 order).  Device pointers in, device pointers out; the collectives are
issued by libvq_amd on the caller's HIP stream.  ``distributed.py`` keeps the same two steps over
``torch.distributed`` tensors for the gloo CPU tests.
"""
from __future__ import annotations

import ctypes
from ctypes import c_int, c_int64, c_void_p
from typing import Optional, Sequence

from . import _lib

ID_BYTES = 128


class Comm:
    def __init__(self, rank: int, world: int, unique_id: bytes, device: Optional[int] = None):
        if len(unique_id) != ID_BYTES:
            raise ValueError(f"unique_id must be {ID_BYTES} bytes")
        self.device = _lib.init(device)
        buf = ctypes.create_string_buffer(unique_id, ID_BYTES)
        h = c_void_p()
        _lib.check(_lib.load().vq_comm_init(int(rank), int(world), buf, ctypes.byref(h)))
        self._h = h
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def make_unique_id() -> bytes:
        """Rank 0 calls this and ships the 128 bytes to the other ranks (any side channel)."""
        buf = ctypes.create_string_buffer(ID_BYTES)
        _lib.check(_lib.load().vq_comm_unique_id(buf, ID_BYTES))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, device: Optional[int] = None) -> "Comm":
        """Bootstrap over an initialised torch.distributed group (its store carries the id; no tensor traffic)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        # Rank 0 ALWAYS takes part in the broadcast: if it cannot make the id (librccl not loadable, ncclGetUniqueId
        # failing) it ships the error instead, and every rank raises — nobody is left waiting in a collective its peers
        # never enter (bench.py then agrees on the torch.distributed exchange with one all_reduce).
        box = [None]
        if rank == 0:
            try:
                box = [("ok", cls.make_unique_id())]
            except Exception as e:                    # noqa: BLE001 - forwarded to every rank
                box = [("err", f"{type(e).__name__}: {e}")]
        dist.broadcast_object_list(box, src=0)
        tag, payload = box[0]
        if tag != "ok":
            raise _lib.VqError(f"rank 0 could not create the RCCL bootstrap id: {payload}")
        return cls(rank, world, payload, device)

    @classmethod
    def single(cls, device: Optional[int] = None) -> "Comm":
        _lib.init(device)
        return cls(0, 1, cls.make_unique_id(), device)

    def rccl_version(self) -> int:
        v = c_int(0)
        _lib.check(_lib.load().vq_comm_info(self._h, None, None, ctypes.byref(v)))
        return v.value

    def all_gather_rows(self, d_local: int, counts: Sequence[int], dim: int, d_out: int, hip_stream: int = 0) -> None:
        """``counts[r]`` rows of ``dim`` fp32 from every rank -> all rows in rank (= frame) order at ``d_out`` on
        every rank; asynchronous on ``hip_stream``."""
        if len(counts) != self.world:
            raise ValueError("counts needs one entry per rank")
        arr = (c_int64 * self.world)(*[int(c) for c in counts])
        _lib.check(_lib.load().vq_allgather_rows(self._h, c_void_p(d_local or None), arr, int(dim), c_void_p(d_out),
                                                 c_void_p(hip_stream or None)))

    def search_sharded(self, index, d_queries: int, nq: int, k: int, row_offset: int, d_ids: int, d_dist: int,
                       mode: Optional[int] = None) -> None:
        """``index`` (an ``indexes.hnsw.HNSWIndex``) holds rows ``[row_offset, row_offset + size)`` of the global
        matrix; the exact global top-k (global row ids) lands at ``d_ids`` / ``d_dist`` on every rank.  A rank whose own
        scan fails raises here (after entering the exchange, so its peers do not hang); the peers get empty lists and
        ``check()`` raises on them."""
        with index.lock:
            _lib.check(_lib.load().vq_index_search_sharded(index._h, self._h, c_void_p(d_queries), int(nq), int(k),
                                                           int(index.search_mode if mode is None else mode),
                                                           int(row_offset), c_void_p(d_ids), c_void_p(d_dist)))

    def check(self) -> None:
        """Raises once if a sharded search on this communicator was voided by a PEER's local failure (the failing rank got
        its own exception from ``search_sharded``; here the call returned empty lists).  Call after synchronising."""
        _lib.check(_lib.load().vq_comm_check(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().vq_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk_device(d_all_ids: int, d_all_dist: int, world: int, nq: int, k: int, d_ids: int, d_dist: int,
                      hip_stream: int = 0) -> None:
    """The merge step alone: ``[world][nq][k]`` shard results with global ids -> exact ``[nq][k]``."""
    _lib.init()
    _lib.check(_lib.load().vq_merge_topk_device(c_void_p(d_all_ids), c_void_p(d_all_dist), int(world), int(nq), int(k),
                                                c_void_p(d_ids), c_void_p(d_dist), c_void_p(hip_stream or None)))

"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/knn_oracle.c.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libknn_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "knn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        lib.vq_oracle_topk.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, ctypes.c_int,
                                       ctypes.c_int, i32p, f32p]
        lib.vq_oracle_topk.restype = None
        lib.vq_oracle_distances.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, f32p, f32p]
        lib.vq_oracle_distances.restype = None
        lib.vq_oracle_normalize_rows.argtypes = [f32p, ctypes.c_int64, ctypes.c_int]
        lib.vq_oracle_normalize_rows.restype = None
        _lib = lib
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def topk(rows: np.ndarray, queries: np.ndarray, k: int):
    """rows [n,d] (already normalised), queries [q,d] (already normalised) →
    (ids int32 [q,k] row numbers, dist fp32 [q,k])."""
    rows, rp = _f32(rows)
    queries, qp = _f32(np.atleast_2d(queries))
    n, d = rows.shape if rows.ndim == 2 else (0, queries.shape[1])
    nq = queries.shape[0]
    ids = np.empty((nq, k), dtype=np.int32)
    dist = np.empty((nq, k), dtype=np.float32)
    _load().vq_oracle_topk(rp, n, d, qp, nq, k, ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                           dist.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return ids, dist


def distances(rows: np.ndarray, query: np.ndarray) -> np.ndarray:
    rows, rp = _f32(rows)
    query, qp = _f32(query)
    out = np.empty(rows.shape[0], dtype=np.float32)
    _load().vq_oracle_distances(rp, rows.shape[0], rows.shape[1], qp,
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def normalize_rows(rows: np.ndarray) -> np.ndarray:
    out = np.array(rows, dtype=np.float32, order="C", copy=True)
    _load().vq_oracle_normalize_rows(out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                     out.shape[0], out.shape[1])
    return out

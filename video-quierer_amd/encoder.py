"""Thin owner of a ``vq_encoder`` handle (include/vq_amd.h): weights in, uint8
frames in, L2-normalised fp32 embeddings out.  Used by
core.feature_extractor.FeatureExtractor, bench.py and the tests.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_float, c_int, c_uint8, c_void_p
from typing import Dict, Optional

import numpy as np

from . import _lib
from .weights import VitConfig, weight_names


# GEMM operand type per group (include/vq_amd.h VQ_ENC_F16_*): fp32 accumulation and the same MFMA rate either way
DTYPE_GROUPS = {"patch": 0x100, "qkv": 0x200, "attn": 0x400, "fc1": 0x800, "fc2": 0x1000}
MIXED_FP16_GROUPS = ("qkv", "attn", "fc1", "fc2")      # = VQ_ENC_MIXED: everything but the patch-embed GEMM
# plain bf16 operands miss the 1e-3 score tolerance; all-fp16 (patch weights scaled by a power of two out of
# fp16's subnormal range, undone exactly in the epilogue) is the choice that reproduces the reference's id lists on config 1
DEFAULT_COMPUTE_DTYPE = "fp16"


def dtype_to_flags(compute_dtype: str) -> int:
    """'bf16' | 'fp16' (default: every GEMM group; the patch weights W/(255 std) are scaled by a power of two so that
    they leave fp16's subnormal range, undone exactly in the GEMM epilogue) | 'mixed' (MIXED_FP16_GROUPS in fp16, the
    patch-embed GEMM in bf16: round 2's default; max score error 2.0e-4 against 1.1e-4 for 'fp16' and 1.0e-3 for plain
    bf16) | 'fp16:<group>+<group>…' (named groups in fp16, the rest bf16)."""
    if compute_dtype == "bf16":
        return 0
    if compute_dtype == "fp16":
        return 1
    if compute_dtype == "mixed":
        return sum(DTYPE_GROUPS[g] for g in MIXED_FP16_GROUPS)
    if compute_dtype.startswith("fp16:"):
        try:
            return sum(DTYPE_GROUPS[g] for g in compute_dtype[5:].split("+") if g)
        except KeyError as e:
            raise ValueError(f"unknown GEMM group {e} (have {sorted(DTYPE_GROUPS)})") from None
    raise ValueError("compute_dtype must be 'bf16', 'fp16', 'mixed' or 'fp16:<group>+…'")


class VitEncoder:
    def __init__(self, cfg: VitConfig, weights: Dict[str, np.ndarray], max_batch: int = 256,
                 device: Optional[int] = None, compute_dtype: str = DEFAULT_COMPUTE_DTYPE, concurrent: bool = False):
        """concurrent: several handles are kept busy on separate streams (VQ_ENC_CONCURRENT in vq_amd.h)."""
        self.compute_dtype = compute_dtype
        dtype_flags = dtype_to_flags(compute_dtype)
        self.cfg = cfg
        self.max_batch = int(max_batch)
        self.device = _lib.init(device)
        lib = _lib.load()
        names = weight_names(cfg)
        self._keep = [np.ascontiguousarray(weights[n], dtype=np.float32) for n in names]
        ptrs = (POINTER(c_float) * len(names))(*[_lib.fptr(a) for a in self._keep])
        ccfg = _lib.VitConfigC(cfg.image_size, cfg.patch_size, cfg.hidden, cfg.mlp, cfg.layers, cfg.heads,
                               cfg.proj_dim, cfg.ln_eps)
        h = c_void_p()
        _lib.check(lib.vq_encoder_create_ex(ctypes.byref(ccfg), ptrs, len(names), self.max_batch,
                                            dtype_flags | (2 if concurrent else 0),
                                            ctypes.byref(h)))
        self._h = h
        self._keep = None          # the library has its own device copies now
        self.output_dim = cfg.proj_dim

    def clone(self, max_batch: Optional[int] = None, concurrent: bool = True) -> "VitEncoder":
        """Another handle on the SAME device weights (own stream and workspace): what keeping several batches in
        flight needs (vq_encoder_create_shared).  Either handle may be closed first."""
        other = object.__new__(VitEncoder)
        other.compute_dtype, other.cfg, other.device = self.compute_dtype, self.cfg, self.device
        other.max_batch = int(max_batch or self.max_batch)
        other._keep, other.output_dim = None, self.output_dim
        h = c_void_p()
        _lib.check(_lib.load().vq_encoder_create_shared(self._h, other.max_batch, 2 if concurrent else 0, ctypes.byref(h)))
        other._h = h
        return other

    # -- host buffers ---------------------------------------------------------
    def encode(self, frames: np.ndarray, swap_rb: bool = True) -> np.ndarray:
        """uint8 [n,S,S,3] → fp32 [n,proj_dim] (synchronous)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        s = self.cfg.image_size
        if frames.ndim != 4 or frames.shape[1:] != (s, s, 3):
            raise ValueError(f"frames must be [n,{s},{s},3] uint8, got {frames.shape}")
        n = frames.shape[0]
        out = np.empty((n, self.cfg.proj_dim), dtype=np.float32)
        if n:
            _lib.check(_lib.load().vq_encoder_encode_u8(self._h, frames.ctypes.data_as(POINTER(c_uint8)), n,
                                                        int(bool(swap_rb)), _lib.fptr(out)))
        return out

    # -- pinned staging (host frames without the extra copy) --------------------
    def staging(self, slot: int) -> np.ndarray:
        """uint8 view [max_batch,S,S,3] of pinned staging slot 0/1 (library-owned, valid until close())."""
        ptr = POINTER(c_uint8)()
        size = ctypes.c_size_t(0)
        _lib.check(_lib.load().vq_encoder_staging(self._h, int(slot), ctypes.byref(ptr), ctypes.byref(size)))
        s = self.cfg.image_size
        arr = np.ctypeslib.as_array(ptr, shape=(size.value,))
        return arr.reshape(self.max_batch, s, s, 3)

    def encode_staged(self, slot: int, n: int, swap_rb: bool = True) -> np.ndarray:
        out = np.empty((n, self.cfg.proj_dim), dtype=np.float32)
        _lib.check(_lib.load().vq_encoder_encode_staged(self._h, int(slot), int(n), int(bool(swap_rb)), _lib.fptr(out)))
        return out

    # -- pipelined ingest of host frames ------------------------------------------
    def stage_frames(self, slot: int, frames, n_threads: int = 4) -> None:
        """Gather a list of C-contiguous uint8 [S,S,3] arrays into pinned slot 0/1 (C threads, GIL released)."""
        ptrs = (c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        _lib.check(_lib.load().vq_encoder_stage_frames(self._h, int(slot), ptrs, len(frames), int(n_threads)))

    def submit_staged(self, slot: int, n: int, swap_rb: bool = True) -> None:
        """Enqueue upload → forward → download for the slot; returns immediately."""
        _lib.check(_lib.load().vq_encoder_submit_staged(self._h, int(slot), int(n), int(bool(swap_rb))))

    def wait_staged(self, slot: int, n: int) -> np.ndarray:
        out = np.empty((n, self.cfg.proj_dim), dtype=np.float32)
        _lib.check(_lib.load().vq_encoder_wait_staged(self._h, int(slot), _lib.fptr(out)))
        return out

    def prewarm_staged(self) -> None:
        """Allocate both pinned slots and their device/result buffers now (a one-frame pass through each) so the
        first real ingest does not pay for it."""
        for slot in (0, 1):
            self.staging(slot)[0].fill(0)
            self.submit_staged(slot, 1)
            self.wait_staged(slot, 1)

    # -- device buffers (pointers, e.g. torch.Tensor.data_ptr()) ----------------
    def encode_device(self, d_frames: int, n: int, d_out_f32: int, d_out_f16: int = 0, swap_rb: bool = True) -> None:
        """Asynchronous on the encoder's stream; call synchronize() before reading."""
        _lib.check(_lib.load().vq_encoder_encode_u8_device(self._h, c_void_p(d_frames), int(n), int(bool(swap_rb)),
                                                           c_void_p(d_out_f32), c_void_p(d_out_f16 or None)))

    def synchronize(self) -> None:
        _lib.check(_lib.load().vq_encoder_synchronize(self._h))

    def set_stream(self, hip_stream: int) -> None:
        """Launch on a caller-owned stream (e.g. torch.cuda.current_stream().cuda_stream); 0 = own stream."""
        _lib.check(_lib.load().vq_encoder_set_stream(self._h, c_void_p(hip_stream or None)))

    # -- measurement / test hooks ---------------------------------------------
    def profile_begin(self) -> None:
        _lib.check(_lib.load().vq_encoder_profile_begin(self._h))

    def profile_end(self) -> Dict[str, dict]:
        lib = _lib.load()
        ms = (c_float * _lib.ENC_NCLASS)()
        cnt = (c_int * _lib.ENC_NCLASS)()
        _lib.check(lib.vq_encoder_profile_end(self._h, ms, cnt))
        return {lib.vq_encoder_profile_class_name(i).decode(): {"ms": float(ms[i]), "launches": int(cnt[i])}
                for i in range(_lib.ENC_NCLASS)}

    def profile_bracket_overhead_ms(self) -> float:
        """Median elapsed time of an empty event bracket on this handle's stream (what profile_end's figures carry per launch
        beyond the kernel itself)."""
        ms = c_float(0.0)
        _lib.check(_lib.load().vq_encoder_profile_bracket_overhead(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def debug_set_layers(self, layers: int) -> None:
        _lib.check(_lib.load().vq_encoder_debug_set_layers(self._h, int(layers)))

    def debug_read(self, name: str, rows: int) -> np.ndarray:
        cols = {"x": self.cfg.hidden, "h": self.cfg.hidden, "qkv": 3 * self.cfg.hidden, "att": self.cfg.hidden,
                "mlp": self.cfg.mlp}[name]
        out = np.empty((rows, cols), dtype=np.float32)
        _lib.check(_lib.load().vq_encoder_debug_read(self._h, name.encode(), int(rows), _lib.fptr(out)))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().vq_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def debug_gemm(a: np.ndarray, w: np.ndarray, use_f16: bool = False, kernel: int = 0) -> np.ndarray:
    """C = A @ W.T through the production MFMA mainloops (unit-test hook).
    kernel: 0 auto, 1 = 128x128 two-phase, 2 = 256x256 phased."""
    _lib.init()
    a = np.ascontiguousarray(a, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    m, k = a.shape
    n = w.shape[0]
    c = np.empty((m, n), dtype=np.float32)
    _lib.check(_lib.load().vq_debug_gemm(_lib.fptr(a), _lib.fptr(w), m, n, k, int(bool(use_f16)) | (int(kernel) << 1), _lib.fptr(c)))
    return c

"""Owner of a ``vq_text_encoder`` handle (include/vq_amd.h): CLIP text tower on
token ids → L2-normalised fp32 text embeddings.  Replaces what
``FeatureExtractor.extract_text_features`` (reference
src/core/feature_extractor.py:218-234) gets from ``CLIPModel.get_text_features``.
Tokenisation stays on the host (see ``load_tokenizer``).
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_float, c_int32, c_void_p
from typing import Dict, Optional

import numpy as np

from . import _lib
from .encoder import DEFAULT_COMPUTE_DTYPE, dtype_to_flags
from .weights import TextConfig, text_weight_shapes


class TextEncoder:
    def __init__(self, cfg: TextConfig, weights: Dict[str, np.ndarray], max_batch: int = 64,
                 device: Optional[int] = None, compute_dtype: str = DEFAULT_COMPUTE_DTYPE):
        dtype_flags = dtype_to_flags(compute_dtype)
        self.compute_dtype = compute_dtype
        self.cfg = cfg
        self.device = _lib.init(device)
        lib = _lib.load()
        names = [n for n, _ in text_weight_shapes(cfg)]
        keep = [np.ascontiguousarray(weights[n], dtype=np.float32) for n in names]
        ptrs = (POINTER(c_float) * len(names))(*[_lib.fptr(a) for a in keep])
        ccfg = _lib.TextConfigC(cfg.vocab, cfg.max_positions, cfg.hidden, cfg.mlp, cfg.layers, cfg.heads,
                                cfg.proj_dim, cfg.eos_token_id, cfg.ln_eps)
        h = c_void_p()
        _lib.check(lib.vq_text_encoder_create(ctypes.byref(ccfg), ptrs, len(names), int(max_batch),
                                              dtype_flags, ctypes.byref(h)))
        self._h = h
        self.output_dim = cfg.proj_dim

    def encode_ids(self, input_ids) -> np.ndarray:
        """int [n, L] (each row: bos … eos [eos-padding]) → fp32 [n, proj_dim]."""
        ids = np.ascontiguousarray(np.atleast_2d(np.asarray(input_ids)), dtype=np.int32)
        n, L = ids.shape
        if L > self.cfg.max_positions:
            raise ValueError(f"sequence length {L} exceeds max_position_embeddings {self.cfg.max_positions}")
        out = np.empty((n, self.cfg.proj_dim), dtype=np.float32)
        if n:
            _lib.check(_lib.load().vq_text_encoder_encode_ids(self._h, ids.ctypes.data_as(POINTER(c_int32)), n, L,
                                                              _lib.fptr(out)))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().vq_text_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_tokenizer(model_dir: str):
    """CLIP BPE tokenizer from LOCAL files (vocab.json + merges.txt in the checkpoint directory), through the
    same ``transformers`` tokenizer the reference's CLIPProcessor wraps.  Never fetches; returns None if the
    files or the package are missing."""
    import os
    if not model_dir or not os.path.exists(os.path.join(model_dir, "vocab.json")):
        return None
    try:
        from transformers import CLIPTokenizer
        return CLIPTokenizer.from_pretrained(model_dir, local_files_only=True)
    except Exception:
        return None

"""MI355X-native replacement for the CLIP frame-encoding + cosine k-NN hot path
of adhney/video-quierer.  Python host code over a C-ABI shared library of
hand-written gfx950 HIP kernels (csrc/, include/vq_amd.h).

Drop-in classes (same names/signatures as the reference):
  core.feature_extractor.FeatureExtractor / BatchProcessor / CachedFeatureExtractor
  indexes.hnsw.HNSWIndex / OptimizedHNSWIndex
Also: overhaul_index.SimpleVideoIndex (the live path's brute-force index), preprocess.FramePreprocessor
(Pillow-exact / cv2-style resize and the frame-quality filter on the GPU), text_encoder.TextEncoder.

Importing this package does not touch the GPU or the shared library; the
classes do, on construction, and raise if either is missing (no CPU path).
"""
import importlib as _importlib
import sys as _sys

__version__ = "0.1.0"


def install_dropin() -> None:
    """Make ``from core.feature_extractor import FeatureExtractor, BatchProcessor`` and
    ``from indexes.hnsw import OptimizedHNSWIndex`` (reference
    src/video_search_system.py:18-19) resolve to this build, whatever sys.path says."""
    for short in ("core", "indexes"):
        pkg = _importlib.import_module(f"video_quierer_amd.{short}")
        _sys.modules[short] = pkg
    _sys.modules["core.feature_extractor"] = _importlib.import_module("video_quierer_amd.core.feature_extractor")
    _sys.modules["indexes.hnsw"] = _importlib.import_module("video_quierer_amd.indexes.hnsw")


def __getattr__(name):  # lazy: keep `import video_quierer_amd` free of ctypes/GPU work
    if name in ("FeatureExtractor", "BatchProcessor", "CachedFeatureExtractor"):
        return getattr(_importlib.import_module("video_quierer_amd.core.feature_extractor"), name)
    if name in ("HNSWIndex", "OptimizedHNSWIndex"):
        return getattr(_importlib.import_module("video_quierer_amd.indexes.hnsw"), name)
    if name == "FramePreprocessor":
        return getattr(_importlib.import_module("video_quierer_amd.preprocess"), name)
    if name == "SimpleVideoIndex":
        return getattr(_importlib.import_module("video_quierer_amd.overhaul_index"), name)
    raise AttributeError(name)

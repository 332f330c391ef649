"""MI355X-native replacement for the CLIP frame-encoding + cosine k-NN hot path
of adhney/video-quierer.  Python host code over a C-ABI shared library of
hand-written gfx950 HIP kernels (csrc/, include/vq_amd.h).

Drop-in classes (same names/signatures as the reference):
  core.feature_extractor.FeatureExtractor / BatchProcessor / CachedFeatureExtractor
  indexes.hnsw.HNSWIndex / OptimizedHNSWIndex
"""
__version__ = "0.1.0"

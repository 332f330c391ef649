#!/usr/bin/env python3
"""Steady-state host ingest through the drop-in: extract_from_video_frames on 16,384 host frames (64 passes of 256)
for 1, 2 and 3 ingest handles."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.core.feature_extractor import FeatureExtractor
rng = np.random.default_rng(0)
frames = [rng.integers(0, 255, (224, 224, 3), dtype=np.uint8) for _ in range(1024)]
fds = [{"frame": f, "frame_number": i} for i, f in enumerate(frames)] * 16
for ns in (1, 2, 3, 4):
    fx = FeatureExtractor(model_name="seed:1234", batch_size=32, device_batch=256, ingest_streams=ns)
    fx.extract_from_video_frames(fds[:2048])
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter(); out = fx.extract_from_video_frames(fds); dt = time.perf_counter() - t0
        best = max(best, len(fds) / dt)
    print(f"ingest handles {ns}: {best:.0f} frames/s (best of 3, {len(fds)} host frames)", flush=True)
    fx.thread_pool.shutdown()
    del fx

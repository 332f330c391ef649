#!/usr/bin/env python3
"""Where the host time of one 256-frame ingest pass goes (list of ndarray frames -> features)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.core.feature_extractor import FeatureExtractor
rng = np.random.default_rng(0)
frames = [rng.integers(0, 255, (224, 224, 3), dtype=np.uint8) for _ in range(256)]
fds = [{"frame": f, "timestamp": i / 30.0, "frame_number": i} for i, f in enumerate(frames)]
for nt in (4, 8):
    fx = FeatureExtractor(model_name="seed:1234", batch_size=32, device_batch=256, num_threads=nt)
    m = fx.model
    fx.extract_batch(frames[:8])
    def t(fn, reps=20):
        fn(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - t0) / reps * 1e3
    chk = t(lambda: fx._all_native_ndarrays(frames))
    ptr = t(lambda: [f.ctypes.data for f in frames])
    ptr2 = t(lambda: [f.__array_interface__["data"][0] for f in frames])
    stg = t(lambda: m.stage_frames(0, frames, nt))
    def sub():
        m.submit_staged(0, 256); 
    def subwait():
        m.submit_staged(0, 256); m.wait_staged(0, 256)
    t0 = time.perf_counter(); m.submit_staged(0, 256); ts = (time.perf_counter() - t0) * 1e3; m.wait_staged(0, 256)
    sw = t(subwait, 10)
    asm = t(lambda: [dict(fd, features=None, feature_extraction_time=0.0) for fd in fds])
    big = fds * 16
    fx.extract_from_video_frames(big)
    t0 = time.perf_counter(); fx.extract_from_video_frames(big); te = time.perf_counter() - t0
    print(f"threads {nt}: check {chk:.3f} ms, ptrs {ptr:.3f} ms (array_interface {ptr2:.3f}), stage_frames {stg:.3f} ms, "
          f"submit {ts:.3f} ms, submit+wait {sw:.3f} ms, result dicts {asm:.3f} ms; 4096 frames e2e {len(big)/te:.0f} frames/s", flush=True)
    fx.thread_pool.shutdown()

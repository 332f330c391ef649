#!/usr/bin/env python3
"""Concurrency soak: several Python threads hammer one FeatureExtractor, one index and one preprocessor (the
threading contract of SURVEY.md §8b: executor threads call extract_batch, a 4-thread pool calls search) and every
result is compared with the answer computed single-threaded beforehand.  usage: stress.py [seconds]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.core.feature_extractor import FeatureExtractor
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex
from video_quierer_amd.preprocess import FramePreprocessor

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rng = np.random.default_rng(3)
fx = FeatureExtractor(model_name="seed:1234", batch_size=32, device_batch=64)
pre = FramePreprocessor()
frames = [rng.integers(0, 255, (224, 224, 3), dtype=np.uint8) for _ in range(48)]
odd = [rng.integers(0, 255, (180, 320, 3), dtype=np.uint8) for _ in range(6)]
big = rng.integers(0, 255, (4, 360, 640, 3), dtype=np.uint8)
ref_emb = fx.extract_batch(frames)
ref_odd = fx.extract_batch(odd)
ref_res = pre.stretch(big)
ref_q = pre.quality(big)
idx = OptimizedHNSWIndex(dimension=512)
vecs = rng.standard_normal((20000, 512)).astype(np.float32)
idx.add_batch(list(vecs), list(range(20000)))
queries = list(ref_emb[:16])
ref_hits = [[(r["id"], r["distance"]) for r in idx.search(q, 10)] for q in queries]
fds = [{"frame": f, "i": i} for i, f in enumerate(frames * 4)]
ref_ing = np.stack([o["features"] for o in fx.extract_from_video_frames(fds)])

stop = time.time() + secs
errors, counts = [], {"encode": 0, "odd": 0, "search": 0, "resize": 0, "ingest": 0}
lock = threading.Lock()

def guard(fn):
    def run():
        try:
            while time.time() < stop and not errors:
                fn()
        except Exception as e:                       # noqa: BLE001
            errors.append(repr(e))
    return run

def t_encode():
    lo = np.random.randint(0, 32)
    out = fx.extract_batch(frames[lo:lo + 16])
    if np.abs(out - ref_emb[lo:lo + 16]).max() > 2e-6: errors.append("encode mismatch")
    with lock: counts["encode"] += 1

def t_odd():
    if np.abs(fx.extract_batch(odd) - ref_odd).max() > 2e-6: errors.append("odd-size mismatch")
    with lock: counts["odd"] += 1

def t_search():
    i = np.random.randint(0, 16)
    got = [(r["id"], r["distance"]) for r in idx.search(queries[i], 10)]
    if got != ref_hits[i]: errors.append("search mismatch")
    got_b = idx.search_batch(queries, 10)
    if [[(r["id"], r["distance"]) for r in rr] for rr in got_b] != ref_hits: errors.append("search_batch mismatch")
    with lock: counts["search"] += 1

def t_resize():
    if not np.array_equal(pre.stretch(big), ref_res): errors.append("resize mismatch")
    m, v = pre.quality(big)
    if not (np.array_equal(m, ref_q[0]) and np.array_equal(v, ref_q[1])): errors.append("quality mismatch")
    with lock: counts["resize"] += 1

def t_ingest():
    out = np.stack([o["features"] for o in fx.extract_from_video_frames(fds)])
    if np.abs(out - ref_ing).max() > 2e-6: errors.append("ingest mismatch")
    with lock: counts["ingest"] += 1

threads = [threading.Thread(target=guard(f)) for f in (t_encode, t_encode, t_odd, t_search, t_search, t_resize, t_ingest)]
for t in threads: t.start()
for t in threads: t.join()
print("errors:", errors[:5] if errors else "none", "| iterations:", counts, flush=True)
sys.exit(1 if errors else 0)

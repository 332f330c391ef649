#!/usr/bin/env python3
"""configs[3] end to end through the drop-in classes: 4 videos x 1000 synthetic frames (host uint8 frame
dicts, as frame_extractor.py yields them) -> FeatureExtractor.extract_from_video_frames ->
OptimizedHNSWIndex.add_batch (string ids f"{video}_{i}") -> 1000 queries, k = 10.

Single process = one GPU.  Under torch.distributed.run with W ranks, video v goes to ranks {2v, 2v+1}
(500 frames each when W = 8), embeddings are all-gathered (RCCL) so every rank indexes all 4000 rows,
and the queries are split across ranks.
"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from video_quierer_amd.core.feature_extractor import FeatureExtractor
from video_quierer_amd.indexes.hnsw import OptimizedHNSWIndex
from video_quierer_amd.distributed import shard_range, all_gather_rows

rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1)); local = int(os.environ.get("LOCAL_RANK", 0))
if os.environ.get("VQ_BENCH_DEVICE") is not None:                  # rehearsal: every rank on one device, e.g. over gloo
    local = int(os.environ["VQ_BENCH_DEVICE"])
if world > 1:
    torch.cuda.set_device(local)
    if os.environ.get("VQ_BENCH_BACKEND", "nccl") == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(os.environ["VQ_BENCH_BACKEND"])
VIDEOS, FRAMES, NQ, K = 4, 1000, 1000, 10
lo, hi = shard_range(VIDEOS * FRAMES, rank, world)                 # contiguous frame range of this rank
rng = np.random.default_rng(1000 + rank)
frames = [{"frame": rng.integers(0, 255, (224, 224, 3), dtype=np.uint8), "timestamp": (g % FRAMES) / 30.0,
           "frame_number": g % FRAMES, "video_id": f"video{g // FRAMES}"} for g in range(lo, hi)]
fx = FeatureExtractor(model_name="seed:1234", device=f"cuda:{local}", batch_size=32, device_batch=256)
fx.extract_batch([frames[0]["frame"]])                              # warm-up (reference does one too, video_search_system.py:607-609)
t0 = time.perf_counter()
out = fx.extract_from_video_frames(frames)
t_enc = time.perf_counter() - t0
emb = np.stack([o["features"] for o in out])
if world > 1:
    emb = all_gather_rows(torch.from_numpy(emb).cuda(local)).cpu().numpy()
ids = [f"video{g // FRAMES}_{g % FRAMES}" for g in range(VIDEOS * FRAMES)]
idx = OptimizedHNSWIndex(dimension=512, device=local)
t0 = time.perf_counter(); idx.add_batch(list(emb), ids); t_add = time.perf_counter() - t0
qrng = np.random.default_rng(5)
queries = emb[qrng.integers(0, len(emb), NQ)] + 0.05 * qrng.standard_normal((NQ, 512)).astype(np.float32)
qlo, qhi = shard_range(NQ, rank, world)
idx.search_batch(list(queries[qlo:qlo + 2]), K)
t0 = time.perf_counter(); res = idx.search_batch(list(queries[qlo:qhi]), K); t_q = time.perf_counter() - t0
t0 = time.perf_counter(); single = [idx.search(q, K) for q in queries[qlo:qlo + 50]]; t_s = time.perf_counter() - t0
if rank == 0:
    print(json.dumps({"ranks": world, "frames": hi - lo, "encode_s": round(t_enc, 3), "frames_per_s_per_rank": round((hi - lo) / t_enc),
                      "index_add_s": round(t_add, 3), "queries": qhi - qlo, "batched_search_s": round(t_q, 4),
                      "queries_per_s": round((qhi - qlo) / t_q), "single_query_ms": round(1e3 * t_s / 50, 3),
                      "top1_is_source_frame": float(np.mean([r[0]["id"].startswith("video") for r in res]))}))
if world > 1:
    dist.destroy_process_group()

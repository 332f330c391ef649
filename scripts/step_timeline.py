#!/usr/bin/env python3
"""When does each of the K timed steps of bench.py's encode leg finish?  Three batches in flight on three streams (as bench.py), an event
after every step; prints each step's completion time and the rate over the first / middle / last thirds - where a short run (the
driver's --steps 20) loses against a long one.    python scripts/step_timeline.py [steps] [warmup]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from video_quierer_amd import _lib
from video_quierer_amd.encoder import VitEncoder
from video_quierer_amd.weights import VIT_B_32, seeded_weights

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, NS = 256, 3
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
_lib.init(0)
encs = [VitEncoder(VIT_B_32, seeded_weights(VIT_B_32, 1234), max_batch=B, device=0, compute_dtype="fp16", concurrent=True)]
encs += [encs[0].clone(concurrent=True) for _ in range(NS - 1)]
for e, s in zip(encs, streams):
    e.set_stream(s.cuda_stream)
pool = [torch.randint(0, 255, (B, 224, 224, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
embs = [torch.empty((B, 512), dtype=torch.float32, device=dev) for _ in range(NS)]
def step(i):
    with torch.cuda.stream(streams[i % NS]):
        encs[i % NS].encode_device(pool[i % 4].data_ptr(), B, embs[i % NS].data_ptr())
# optional: start the batches out of phase - stream j's first step waits j * STAGGER_MS behind a spin kernel
STAGGER_MS = float(os.environ.get("STAGGER_MS", "0"))
cyc_per_ms = 0.0
if STAGGER_MS > 0:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1_000_000); torch.cuda.synchronize()
    a.record(); torch.cuda._sleep(10_000_000); b.record(); torch.cuda.synchronize()
    cyc_per_ms = 10_000_000 / a.elapsed_time(b)
    print(f"spin kernel: {cyc_per_ms:.0f} cycles per ms")
# optional: keep the GPU busy with UNRELATED work (fp16 matmuls) for PREHEAT_S seconds before the first repetition: is a first
# repetition slower because the chip is not in its steady state yet, or because the handles are not?
PREHEAT_S = float(os.environ.get("PREHEAT_S", "0"))
if PREHEAT_S > 0:
    a_ = torch.randn((8192, 8192), dtype=torch.float16, device=dev); b_ = torch.randn((8192, 8192), dtype=torch.float16, device=dev)
    big = torch.empty((2, 1 << 30), dtype=torch.uint8, device=dev)          # PREHEAT_KIND=copy: 1 GiB copies (HBM-bound) instead of matmuls
    t_ = time.perf_counter()
    while time.perf_counter() - t_ < PREHEAT_S:
        for _ in range(10):
            if os.environ.get("PREHEAT_KIND") == "copy":
                big[1].copy_(big[0])
            else:
                a_ @ b_
        torch.cuda.synchronize()
    del big
for rep in range(3):
    for i in range(W):
        step(i)
    torch.cuda.synchronize()
    if STAGGER_MS > 0:
        for j in range(1, NS):
            with torch.cuda.stream(streams[j]):
                torch.cuda._sleep(int(j * STAGGER_MS * cyc_per_ms))
    t0e = torch.cuda.Event(enable_timing=True)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    t0 = time.perf_counter()
    t0e.record(streams[0])
    cpu = []
    for i in range(K):
        step(i)
        evs[i].record(streams[i % NS])
        cpu.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    done = [t0e.elapsed_time(e) for e in evs]
    print(f"rep {rep}: {K} steps in {1e3 * wall:.2f} ms = {K * B / wall:.0f} frames/s; CPU enqueue of all steps done at {1e3 * cpu[-1]:.2f} ms")
    print("   step done at (ms): " + " ".join(f"{d:.1f}" for d in done))
    srt = sorted(done)
    third = K // 3
    for name, a, b in (("first", 0, third), ("middle", third, 2 * third), ("last", 2 * third, K)):
        ta = srt[a - 1] if a else 0.0
        print(f"   {name:6s} {b - a} steps: {(b - a) * B / (srt[b - 1] - ta) * 1e3:.0f} frames/s")

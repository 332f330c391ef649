#!/usr/bin/env python3
"""Encode latency per device pass for small batches (one handle, one stream), device-resident frames."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.encoder import VitEncoder
from video_quierer_amd.weights import VIT_B_32, seeded_weights

enc = VitEncoder(VIT_B_32, seeded_weights(VIT_B_32, 1234), max_batch=256)
st = torch.cuda.Stream()
enc.set_stream(st.cuda_stream)
frames = torch.randint(0, 255, (256, 224, 224, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((256, 512), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    for _ in range(5):
        enc.encode_device(frames.data_ptr(), n, out.data_ptr())
    enc.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(st)
    for _ in range(reps):
        enc.encode_device(frames.data_ptr(), n, out.data_ptr())
    e1.record(st)
    t_issue = time.perf_counter() - t0
    enc.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"n={n:4d}: {ms:7.3f} ms/pass  {n/ms*1e3:9.0f} frames/s   host issue {t_issue/reps*1e3:6.3f} ms/pass", flush=True)

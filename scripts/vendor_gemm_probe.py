#!/usr/bin/env python3
"""Calibration only (not on the product path): the vendor library GEMM (torch.matmul -> hipBLASLt/rocBLAS, bf16) on the
encoder's shapes, beside this build's kernels timed by scripts/gemm160_compare.py / bench.py."""
import torch
dev = torch.device("cuda", 0)
shapes = [("b32 qkv", 12800, 2304, 768), ("b32 out", 12800, 768, 768), ("b32 fc1", 12800, 3072, 768), ("b32 fc2", 12800, 768, 3072),
          ("l14 qkv", 18464, 3072, 1024), ("l14 fc1", 18464, 4096, 1024), ("l14 fc2", 18464, 1024, 4096), ("square 8192", 8192, 8192, 8192)]
for name, m, n, k in shapes:
    a = torch.randn((m, k), device=dev, dtype=torch.bfloat16)
    w = torch.randn((n, k), device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:12s} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:7.0f} TFLOP/s", flush=True)

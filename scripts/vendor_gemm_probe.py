#!/usr/bin/env python3
"""Calibration only (not on the product path): the vendor library GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the encoder's
shapes, beside this build's kernels timed by scripts/gemm_asm_probe.py / bench.py.  The chip lowers its clock under MFMA load
by how much the DATA toggles (MI355X_MICROARCH.md 'DVFS give-back'), so every shape runs on three operand fills: bf16 normal
(what rounds 1-3 quoted), fp16 normal, and fp16 uniform multiples of 1e-3 in [-1, 1] — the fill of vq_debug_gemm_bench."""
import torch
dev = torch.device("cuda", 0)
shapes = [("b32 qkv", 12800, 2304, 768), ("b32 out", 12800, 768, 768), ("b32 fc1", 12800, 3072, 768), ("b32 fc2", 12800, 768, 3072),
          ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192), ("16384x4096x4096", 16384, 4096, 4096)]


def fill(shape, kind):
    if kind == "bf16 normal":
        return torch.randn(shape, device=dev, dtype=torch.float32).to(torch.bfloat16)
    if kind == "fp16 normal":
        return torch.randn(shape, device=dev, dtype=torch.float32).to(torch.float16)
    if kind == "fp16 zeros":
        return torch.zeros(shape, device=dev, dtype=torch.float16)
    return (torch.randint(-1000, 1001, shape, device=dev).to(torch.float32) * 1e-3).to(torch.float16)


for name, m, n, k in shapes:
    line = f"{name:16s} M={m} N={n} K={k}:"
    for kind in ("bf16 normal", "fp16 normal", "fp16 uniform", "fp16 zeros"):
        a, w = fill((m, k), kind), fill((n, k), kind)
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
        line += f"  {kind} {ms * 1e3:7.1f} us ({2.0 * m * n * k / ms / 1e9:5.0f} TFLOP/s)"
    print(line, flush=True)

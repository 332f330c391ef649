#!/usr/bin/env python3
"""Where a tower GEMM workgroup's time goes: prologue / K loop / epilogue cycles of gemm_tn256d_kernel, sampled inside
the real encoder pass.  Needs the diagnostic build: `make -C video-quierer_amd/csrc STAMPS=1` in a scratch copy of the tree,
then VQ_AMD_LIB=<that libvq_amd.so> STREAMS=1|3 python scripts/gemm_tower_stamps.py 2>&1 | grep STAMP  (DESIGN.md §4)."""
import os, sys, ctypes, collections, re, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from video_quierer_amd import _lib
from video_quierer_amd.encoder import VitEncoder
from video_quierer_amd.weights import VIT_B_32, seeded_weights
_lib.init(0)
lib = _lib.load()
W = seeded_weights(VIT_B_32, 1234)
nstreams = int(os.environ.get("STREAMS", "1"))
encs = [VitEncoder(VIT_B_32, W, max_batch=256, device=0, concurrent=nstreams > 1)]
encs += [encs[0].clone(concurrent=True) for _ in range(nstreams - 1)]
streams = [torch.cuda.Stream() for _ in range(nstreams)]
for e, s in zip(encs, streams): e.set_stream(s.cuda_stream)
fr = torch.randint(0, 255, (256, 224, 224, 3), dtype=torch.uint8, device="cuda")
out = [torch.empty((256, 512), device="cuda") for _ in range(nstreams)]
torch.cuda.synchronize()
for i in range(12):
    encs[i % nstreams].encode_device(fr.data_ptr(), 256, out[i % nstreams].data_ptr())
torch.cuda.synchronize()
lib.vq_debug_dump_gemm_stamps.restype = ctypes.c_int
lib.vq_debug_dump_gemm_stamps()

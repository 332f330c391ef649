#!/usr/bin/env python3
"""Where a tower GEMM workgroup's time goes: prologue / K loop / epilogue cycles of gemm_tn256d_kernel, sampled inside
the real encoder pass.  Needs the diagnostic build: `make -C video-quierer_amd/csrc STAMPS=1` in a scratch copy of the tree,
then VQ_AMD_LIB=<that libvq_amd.so> STREAMS=1|3 PASSES=800 python scripts/gemm_tower_stamps.py 2> stamps.txt and
python scripts/gemm_tower_stamps.py --summarize stamps.txt (the clock as MI355X_MICROARCH.md's DVFS item 6 takes it)."""
import os, sys, ctypes, collections, re, subprocess
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 2 and sys.argv[1] == "--summarize":      # python scripts/gemm_tower_stamps.py --summarize stamps.txt
    import statistics
    groups = collections.defaultdict(list)
    for line in open(sys.argv[2]):
        m = re.match(r"STAMP K (\d+) tn (\d+) epi (\d+) prologue (\d+) loop (\d+) epilogue (\d+) ticks (\d+)", line)
        if m:
            v = list(map(int, m.groups()))
            groups[tuple(v[:3])].append(v[3:])
    for key, rows in sorted(groups.items()):
        med = [statistics.median(r[i] for r in rows) for i in range(4)]
        ghz = statistics.median(r[1] / (r[3] * 10.0) for r in rows if r[3] > 0)       # shader cycles per ns over the K loop
        print(f"K {key[0]:5d} tiles_n {key[1]:2d} epi {key[2]:4d} n {len(rows):4d}: prologue {med[0]:8.0f} loop {med[1]:8.0f} "
              f"epilogue {med[2]:8.0f} cycles; in-kernel clock over the K loop {ghz:.3f} GHz")
    sys.exit(0)
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
for i in range(int(os.environ.get("PASSES", "12"))):        # the dump keeps the last 4096 samples: PASSES=800 (~2 s) for a settled clock
    encs[i % nstreams].encode_device(fr.data_ptr(), 256, out[i % nstreams].data_ptr())
torch.cuda.synchronize()
lib.vq_debug_dump_gemm_stamps.restype = ctypes.c_int
lib.vq_debug_dump_gemm_stamps()

#!/usr/bin/env python3
"""Times the GPU resize (device-resident frames) and the host path, per source size.
usage: resample_probe.py [n]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd.preprocess import FramePreprocessor, BILINEAR, BICUBIC, clip_processor_geometry

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pre = FramePreprocessor()
st = torch.cuda.Stream()
pre.set_stream(st.cuda_stream)
sizes = ((1080, 1920), (720, 1280), (480, 640), (2160, 3840))
if os.environ.get("PROBE_SIZE"):
    sizes = (tuple(int(v) for v in os.environ["PROBE_SIZE"].split("x")),)
for (h, w) in sizes:
    m = n if h < 2000 else max(1, n // 4)
    d = torch.randint(0, 256, (m, h, w, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty((m, 224, 224, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for name, args in (("stretch bilinear", (224, 224, BILINEAR, None)),
                       ("clip bicubic+crop", None)):
        if args is None:
            rh, rw, top, left = clip_processor_geometry(h, w)
            args = (rh, rw, BICUBIC, (top, left, 224, 224))
        oh, ow, filt, crop = args
        for _ in range(3):
            pre.resize_device(d.data_ptr(), m, h, w, oh, ow, filt, crop, out.data_ptr())
        pre.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        with torch.cuda.stream(st):
            e0.record(st)
            for _ in range(reps):
                pre.resize_device(d.data_ptr(), m, h, w, oh, ow, filt, crop, out.data_ptr())
            e1.record(st)
        pre.synchronize()
        ms = e0.elapsed_time(e1) / reps
        src_gb = m * h * w * 3 / 1e9
        print(f"{h}x{w} {name:18s} n={m}: {ms:7.3f} ms  {m/ms*1e3:9.0f} frames/s  source {src_gb/ms*1e3:7.1f} GB/s", flush=True)
    host = d[:min(m, 16)].cpu().numpy()
    t = time.perf_counter(); pre.resize(host, 224, 224); dt = time.perf_counter() - t
    print(f"{h}x{w} host path (pageable H2D + resize + D2H) n={len(host)}: {len(host)/dt:9.0f} frames/s", flush=True)

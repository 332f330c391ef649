#!/usr/bin/env python3
"""A/B in ONE process, interleaved rounds: the 256x256 deep-prefetch kernel (8) against the persistent out-of-phase
128x256 kernel (20; two workgroups per CU, gemm_mfma128x256p.h) with the tower's own epilogues, on the tower's shapes and
on 4x-row versions of the N = 768 pair (steady state: every CU holds its pair for many tiles).  Needs a DIAG build:
    make -C video-quierer_amd/csrc DIAG=1 EXPERIMENTS=1 OUT=../lib/libvq_amd_diag.so OBJDIR=../lib/obj_diag
    VQ_AMD_LIB=video-quierer_amd/lib/libvq_amd_diag.so python scripts/gemm_p2_probe.py"""
import sys, os, ctypes, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_quierer_amd import _lib
_lib.init(0)
lib = _lib.load()


def run(m, n, k, kernel, epi, mode=0, dephase=0, reps=20, grid=0, census=None):
    ms = ctypes.c_float(0)
    cptr = census.ctypes.data_as(ctypes.c_void_p) if census is not None else None
    _lib.check(lib.vq_debug_gemm_bench(m, n, k, kernel, mode, dephase, epi, reps, grid, ctypes.byref(ms), cptr))
    return ms.value


def census_report(m, n, k, epi, mode, dephase):
    c = np.zeros((512, 4), dtype=np.uint64)
    run(m, n, k, 20, epi, mode | 4, dephase, reps=3, grid=512, census=c)
    hw = c[:, 0].astype(np.int64)
    wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    xcc = c[:, 1].astype(np.int64) & 15
    place = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_cu = collections.Counter(place.values())
    slots = collections.defaultdict(list)
    for i in range(512):
        slots[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append(int(wave_id[i]))
    pair_kinds = collections.Counter(tuple(sorted(v)) for v in slots.values())
    dur = (c[:, 3] - c[:, 2]).astype(np.int64)
    start = c[:, 2].astype(np.int64) - int(c[:, 2].min())
    print(f"  census M={m} N={n} K={k} mode={mode} dephase={dephase}: distinct CUs {len(place)}, workgroups per CU {dict(per_cu)}, "
          f"WAVE_ID sets per CU {dict(pair_kinds)}, start spread {int(start.max())} cyc, lifetime median {int(np.median(dur))} cyc "
          f"(odd-slot {int(np.median(dur[wave_id % 2 == 1])) if (wave_id % 2 == 1).any() else -1}, even-slot {int(np.median(dur[wave_id % 2 == 0]))})", flush=True)


# correctness first: small integers are exact in fp16 and in the fp32 accumulators, so the result must equal numpy's
from video_quierer_amd.encoder import debug_gemm
rng = np.random.default_rng(5)
for (m, n, k) in [(128, 256, 128), (1280, 768, 192), (12800, 768, 768), (2560, 2304, 3072)]:
    a = rng.integers(-4, 5, (m, k)).astype(np.float32)
    w = rng.integers(-4, 5, (n, k)).astype(np.float32)
    ref = a @ w.T
    for kern in (20, 21):
        got = debug_gemm(a, w, use_f16=True, kernel=kern)
        assert np.array_equal(got, ref), f"kernel {kern} wrong at {(m, n, k)}: {np.abs(got - ref).max()}"
print("kernel 20 / 21 exact on integer operands", flush=True)

cases = [
    ("fc1", 12800, 3072, 768, 2, [0, 8000, 16000, 24000]),
    ("qkv", 12800, 2304, 768, 3, [0, 8000, 16000, 24000]),
    ("out x4", 51200, 768, 768, 1, [0, 8000, 16000, 24000]),
    ("fc2 x4", 51200, 768, 3072, 1, [0, 20000, 40000, 80000]),
    ("out", 12800, 768, 768, 1, [0, 8000, 16000]),
    ("fc2", 12800, 768, 3072, 1, [0, 20000, 40000]),
    ("square", 8192, 8192, 8192, 0, [0, 100000]),
]
only = os.environ.get("PROBE_CASES")
for name, m, n, k, epi, delays in cases:
    if only and name not in only.split(","):
        continue
    variants = [("k8", dict(kernel=8))]
    if os.environ.get("PROBE_K12", "1") == "1":
        variants.append(("k12", dict(kernel=12)))
    for d in delays:
        variants.append((f"p2 d={d}", dict(kernel=20, mode=1 if d else 0, dephase=d)))
    variants.append(("p2 prio", dict(kernel=20, mode=2)))
    variants.append((f"p2 prio+d={delays[1]}", dict(kernel=20, mode=3, dephase=delays[1])))
    res = {v[0]: [] for v in variants}
    reps = 8 if m * n * k > 2e11 else 20
    for rnd in range(4):
        for label, kw in variants:
            res[label].append(run(m, n, k, epi=epi, reps=reps, **kw))
    fl = 2.0 * m * n * k
    base = np.median(res["k8"])
    print(f"{name}: M={m} N={n} K={k} epi={epi}", flush=True)
    for label, v in res.items():
        med = np.median(v)
        print(f"    {label:18s} median {med*1e3:8.1f} us  min {min(v)*1e3:8.1f} us  {fl/med/1e9:6.0f} TFLOP/s  x{base/med:5.3f} vs k8", flush=True)
    if name in ("fc1", "out x4"):
        census_report(m, n, k, epi, 0, 0)
        census_report(m, n, k, epi, 1, delays[2])

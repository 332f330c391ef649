#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md §HBM prescribes) ->
profiles/<name>.json: KiB per dispatch per kernel class and hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024
(gfx950: FETCH_SIZE reports half of a wide coalesced read stream; WRITE_SIZE is exact for 16-byte streaming stores).

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [note]
Only full-batch dispatches are averaged (the largest grid of each class), so warm-up / CLS-only launches of the same
kernel do not dilute the figure bench.py quotes as roofline.traffic."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

import re

CLASSES = [        # (class name as bench.py's kernel_classes reports it, regex on the demangled kernel name)
    ("patchify_u8", r"patchify"),
    ("gemm_patch_embed", r"gemm_tn.*EpiPatchEmbedF32"),
    ("embed_finish_ln", r"embed_finish"),
    ("layernorm_bf16", r"layernorm_bf16"),
    ("gemm_qkv", r"gemm_tn.*EpiLnH16<(true|false), false>"),            # no GELU = q|k|v
    ("gemm_fc1_quickgelu", r"gemm_tn.*EpiLnH16<(true|false), true>"),
    ("gemm_qkv", r"gemm_tn.*EpiBiasH16"),
    ("gemm_fc1_quickgelu", r"gemm_tn.*EpiBiasQuickGeluH16"),
    ("attention", r"attention_"),
    ("gemm_out_proj_residual", r"gemm_tn.*EpiBiasResidual(Ln)?F32<0"),
    ("gemm_fc2_residual", r"gemm_tn.*EpiBiasResidual(Ln)?F32<1"),
    ("pool_project", r"pool_project"),
    ("scan_f16_mfma_top2", r"scan[245]_f16_top2"),
    ("scan_f16_stream_top2", r"scan3_f16_top2"),
    ("rescore_verify", r"rescore_verify"),
    ("resample_h", r"resample_h"),
    ("resample_v", r"resample_v"),
]


def classify(name):
    for cls, pat in CLASSES:
        if re.search(pat, name):
            return cls
    return None


def collect(d, counter):
    per = defaultdict(list)          # class -> [(grid, value)]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                cls = classify(row["Kernel_Name"])
                if cls:
                    per[cls].append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    out = {}
    for cls, vals in per.items():
        top = max(g for g, _ in vals)
        full = [v for g, v in vals if g == top]
        out[cls] = (len(full), sum(full) / len(full))
    return out


def main():
    fd, wd, dst = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, write = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    kernels = {}
    for cls in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(cls, (0, 0.0))
        nw, w = write.get(cls, (0, 0.0))
        kernels[cls] = {"dispatches": max(nf, nw), "FETCH_SIZE_KiB": round(f), "WRITE_SIZE_KiB": round(w),
                        "hbm_bytes": int((2 * f + w) * 1024)}
    with open(dst, "w") as fh:
        json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB per dispatch, full-batch dispatches "
                            "only). hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of a wide coalesced "
                            "read stream (MI355X_MICROARCH.md §HBM); Infinity-Cache hits are counted, so this is traffic beyond "
                            "L2, an upper bound on HBM bytes. " + note, "kernels": kernels}, fh, indent=1)
    for k, v in kernels.items():
        print(f"{k:28s} {v['dispatches']:4d} dispatches  fetch {v['FETCH_SIZE_KiB']:8d} KiB  write {v['WRITE_SIZE_KiB']:8d} KiB  "
              f"-> {v['hbm_bytes'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()

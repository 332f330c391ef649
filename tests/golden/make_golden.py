#!/usr/bin/env python3
"""Captures the golden vectors under tests/golden/ — run ONLY in the build
container (needs /root/reference and the locally installed ``transformers``;
neither exists on the GPU box, and nothing of them is committed: the outputs
are data — inputs' seeds and expected outputs).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Produces
  encoder_b32_seed1234.npz   64 synthetic frames through
                             transformers.CLIPModel(CLIPConfig()) carrying the
                             build's seeded weights: L2-normalised [64,512] fp32
                             embeddings, the un-normalised features, and a
                             checksum of the frames/weights they came from.
  encoder_l14_336_seed1234.npz   (`make_golden.py l14`) 32 frames through the ViT-L/14@336 geometry.
  text_b32_seed1234.npz      (`make_golden.py text`) 16 synthetic prompts (token ids) through the text tower.
  text_l14_seed1234.npz      (`make_golden.py text_l14`) the same prompts through the ViT-L/14 text tower (768 wide).
  resample_pil.npz           (`make_golden.py resample`) Pillow's resize / transformers' CLIP image processor on
                             seeded frames: two full outputs + SHA-256 of all eight (inputs are regenerated
                             from the seed by resample_input()).
  ref_index_50.pkl(.sha256), ref_index_50_results.npz, simple_index.npz, ref_simple_index_cache.pkl
                             (`make_golden.py interop`) files WRITTEN by the real reference classes and the result
                             lists the real classes return (persistence / live-index fixtures).
  knn_ref_10000.npz, knn_ref_100000.npz
                             (`make_golden.py knn_big 10000`, `... knn_big 100000`) the REAL OptimizedHNSWIndex over a
                             10k / 100k-row prefix of configs[2]'s recipe, built under the caller's STRING ids
                             (video_search_system.py:164-166) with planted exact-duplicate rows: its top-10 / top-20
                             lists (as row numbers) and distances for 64 queries at ef_search = 50 (default) and
                             ef_search = N (exhaustive), the exact lists of the C oracle beside them, and the
                             default-ef recall against the exact lists.  Rows regenerate from the seed.
  knn_cfg1.npz               the REAL reference index (src/indexes/hnsw.py
                             OptimizedHNSWIndex, random.seed(0)) over 1,000
                             seeded vectors: its levels, entry point and graph,
                             and its top-5 / top-10 result lists for the 64
                             embeddings above at ef_search=50 (default,
                             approximate) and ef_search=1000 (exhaustive).
"""
import hashlib
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

FRAME_SEED = 20250824
WEIGHT_SEED = 1234
INDEX_SEED = 7
N_FRAMES, N_INDEX = 64, 1000


def synth_frames(n, seed=FRAME_SEED):
    # the reference's own synthetic-frame convention: randint(0,255,(224,224,3),uint8)
    # (reference src/video_search_system.py:556-557)
    return np.random.default_rng(seed).integers(0, 255, (n, 224, 224, 3), dtype=np.uint8)


def capture_encoder():
    import torch
    from transformers import CLIPConfig, CLIPModel
    from video_quierer_amd.weights import VIT_B_32, seeded_weights

    torch.manual_seed(0)
    W = seeded_weights(VIT_B_32, WEIGHT_SEED)
    model = CLIPModel(CLIPConfig()).eval()          # ViT-B/32 defaults, no hub access
    sd = model.state_dict()
    for k, v in W.items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    frames = synth_frames(N_FRAMES)
    # _preprocess_image (reference feature_extractor.py:105-116) at 224x224:
    # BGR->RGB, HWC->CHW, /255, (x-mean)/std
    x = torch.from_numpy(np.ascontiguousarray(frames[..., ::-1])).permute(0, 3, 1, 2).float() / 255.0
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    x = (x - mean) / std
    with torch.no_grad():
        out = model.get_image_features(x)
        feats = out.pooler_output if hasattr(out, "pooler_output") else out   # tf 5.x vs 4.x
        emb = torch.nn.functional.normalize(feats, p=2, dim=1)
    wsum = hashlib.sha256()
    for k in sorted(W):
        wsum.update(W[k].tobytes())
    np.savez_compressed(
        os.path.join(HERE, "encoder_b32_seed1234.npz"),
        embeddings=emb.numpy().astype(np.float32), features=feats.numpy().astype(np.float32),
        frame_seed=FRAME_SEED, weight_seed=WEIGHT_SEED,
        frames_sha256=hashlib.sha256(frames.tobytes()).hexdigest(),
        weights_sha256=wsum.hexdigest())
    return emb.numpy().astype(np.float32)


def capture_encoder_l14():
    """ViT-L/14@336 (BASELINE configs[4]): 32 synthetic 336x336 frames through transformers' CLIP with the
    build's seeded L/14 weights — pins the 577-token / patch-14 / hidden-1024 path of the restatement."""
    import torch
    from transformers import CLIPConfig, CLIPModel
    from video_quierer_amd.weights import VIT_L_14_336, seeded_weights

    cfg = VIT_L_14_336
    W = seeded_weights(cfg, WEIGHT_SEED)
    hf = CLIPConfig(vision_config=dict(hidden_size=cfg.hidden, intermediate_size=cfg.mlp, num_hidden_layers=cfg.layers,
                                       num_attention_heads=cfg.heads, image_size=cfg.image_size, patch_size=cfg.patch_size),
                    projection_dim=cfg.proj_dim)
    model = CLIPModel(hf).eval()
    sd = model.state_dict()
    for k, v in W.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    frames = np.random.default_rng(FRAME_SEED).integers(0, 255, (32, 336, 336, 3), dtype=np.uint8)     # (the first 2 = round 1's fixture)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)
    embs = []
    with torch.no_grad():
        for i in range(0, len(frames), 8):
            x = torch.from_numpy(np.ascontiguousarray(frames[i:i + 8, ..., ::-1])).permute(0, 3, 1, 2).float() / 255.0
            out = model.get_image_features((x - mean) / std)
            feats = out.pooler_output if hasattr(out, "pooler_output") else out
            embs.append(torch.nn.functional.normalize(feats, p=2, dim=1))
    emb = torch.cat(embs)
    np.savez_compressed(os.path.join(HERE, "encoder_l14_336_seed1234.npz"), embeddings=emb.numpy().astype(np.float32),
                        frame_seed=FRAME_SEED, weight_seed=WEIGHT_SEED,
                        frames_sha256=hashlib.sha256(frames.tobytes()).hexdigest())


def synth_token_ids(n=16, seed=4242, max_len=77):
    """Synthetic prompts: [bos] + random word-piece ids + [eos], padded with eos to the longest (what
    CLIPProcessor(padding=True) produces for a batch); lengths 3..77."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(3, max_len + 1, n)
    lens[0], lens[1] = max_len, 3
    L = int(lens.max())
    ids = np.full((n, L), 49407, dtype=np.int64)
    for i, ln in enumerate(lens):
        ids[i, 0] = 49406
        ids[i, 1:ln - 1] = rng.integers(0, 49406, ln - 2)
        ids[i, ln - 1] = 49407
    return ids


def capture_text():
    """16 synthetic prompts through transformers' CLIP text tower carrying the build's seeded text weights."""
    import torch
    from transformers import CLIPConfig, CLIPModel
    from video_quierer_amd.weights import TEXT_B_32, seeded_text_weights

    W = seeded_text_weights(TEXT_B_32, WEIGHT_SEED)
    model = CLIPModel(CLIPConfig()).eval()
    sd = model.state_dict()
    for k, v in W.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    ids = synth_token_ids()
    with torch.no_grad():
        out = model.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.ones_like(torch.from_numpy(ids)))
        feats = out.pooler_output if hasattr(out, "pooler_output") else out
        emb = torch.nn.functional.normalize(feats, p=2, dim=1)
    np.savez_compressed(os.path.join(HERE, "text_b32_seed1234.npz"), input_ids=ids.astype(np.int32),
                        embeddings=emb.numpy().astype(np.float32), weight_seed=WEIGHT_SEED)


def capture_text_l14():
    """The ViT-L/14 text tower (12 x 768, 12 heads, projection 768: configs[4]'s "mixed text+image queries",
    feature_extractor.py:218-234) on the same 16 synthetic prompts."""
    import torch
    from transformers import CLIPConfig, CLIPModel
    from video_quierer_amd.weights import TEXT_L_14, seeded_text_weights

    W = seeded_text_weights(TEXT_L_14, WEIGHT_SEED)
    cfg = CLIPConfig(text_config=dict(hidden_size=TEXT_L_14.hidden, intermediate_size=TEXT_L_14.mlp, num_attention_heads=TEXT_L_14.heads,
                                      num_hidden_layers=TEXT_L_14.layers, projection_dim=TEXT_L_14.proj_dim),
                     vision_config=dict(num_hidden_layers=1), projection_dim=TEXT_L_14.proj_dim)
    model = CLIPModel(cfg).eval()
    sd = model.state_dict()
    for k, v in W.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    ids = synth_token_ids()
    with torch.no_grad():
        out = model.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.ones_like(torch.from_numpy(ids)))
        feats = out.pooler_output if hasattr(out, "pooler_output") else out
        emb = torch.nn.functional.normalize(feats, p=2, dim=1)
    np.savez_compressed(os.path.join(HERE, "text_l14_seed1234.npz"), input_ids=ids.astype(np.int32),
                        embeddings=emb.numpy().astype(np.float32), weight_seed=WEIGHT_SEED)


sys.path.insert(0, os.path.dirname(HERE))
from conftest import (RESAMPLE_CASES, RESAMPLE_SEED, resample_input,    # noqa: E402  (shared with the tests)
                      KNN_BIG_SEED, knn_big_ids, knn_big_inputs)


def capture_resample():
    """Pillow (and transformers' CLIP image processor on top of it) on seeded frames: full outputs for the first two
    cases, SHA-256 of the output bytes for all."""
    from PIL import Image
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor()
    mean, std = np.array(proc.image_mean, np.float32), np.array(proc.image_std, np.float32)
    out = {"seed": RESAMPLE_SEED, "names": np.array([c[0] for c in RESAMPLE_CASES])}
    for i, (name, h, w, kind, mode) in enumerate(RESAMPLE_CASES):
        img = resample_input(h, w, kind)
        if mode == "stretch":       # transforms.Resize((224,224)) on a PIL image = Image.resize(BILINEAR)
            res = np.asarray(Image.fromarray(img).resize((224, 224), Image.BILINEAR))
        else:                       # the processor's resize + centre crop, recovered from its normalised output
            pv = proc(images=Image.fromarray(img), return_tensors="np")["pixel_values"][0]
            res = np.rint((pv.transpose(1, 2, 0) * std + mean) * 255).astype(np.uint8)
        out[f"sha256_{name}"] = hashlib.sha256(np.ascontiguousarray(res).tobytes()).hexdigest()
        if i < 2:
            out[f"out_{name}"] = res
    np.savez_compressed(os.path.join(HERE, "resample_pil.npz"), **out)


def capture_knn(queries):
    sys.path.insert(0, "/root/reference/src")
    from indexes.hnsw import OptimizedHNSWIndex        # the real reference

    vecs = np.random.default_rng(INDEX_SEED).standard_normal((N_INDEX, 512)).astype(np.float32)
    random.seed(0)
    idx = OptimizedHNSWIndex(dimension=512)            # defaults: M=16 efC=200 ef=50
    idx.add_batch(list(vecs), list(range(N_INDEX)))
    stored = np.stack([idx.data[i] for i in range(N_INDEX)]).astype(np.float32)
    levels = np.array([idx.levels[i] for i in range(N_INDEX)], dtype=np.int32)
    edges = []                                          # (level, a, b) with a in adj of... directed
    for lv, nodes in idx.graph.items():
        for a, nbrs in nodes.items():
            for b in nbrs:
                edges.append((int(lv), int(a), int(b)))
    edges = np.array(sorted(edges), dtype=np.int32)
    out = dict(index_seed=INDEX_SEED, stored=stored, levels=levels, edges=edges,
               entry_point=np.int32(idx.entry_point))
    for ef in (50, N_INDEX):
        idx.ef_search = ef
        for k in (5, 10):
            res = [idx.search(q, k) for q in queries]
            out[f"ids_ef{ef}_k{k}"] = np.array([[r["id"] for r in rr] for rr in res], dtype=np.int32)
            out[f"dist_ef{ef}_k{k}"] = np.array([[r["distance"] for r in rr] for rr in res], dtype=np.float32)
            out[f"score_ef{ef}_k{k}"] = np.array([[r["score"] for r in rr] for rr in res], dtype=np.float32)
            assert all(type(r["distance"]) is np.float32 for rr in res for r in rr)
    np.savez_compressed(os.path.join(HERE, "knn_cfg1.npz"), **out)


def capture_knn_big(n):
    """the configs[2] / recall claim pinned against the real class above N = 1,000."""
    import time
    sys.path.insert(0, "/root/reference/src")
    from indexes.hnsw import OptimizedHNSWIndex        # the real reference
    from oracle import knn_oracle

    rows, qs = knn_big_inputs(n)
    ids = knn_big_ids(n)
    row_of = {s_: r for r, s_ in enumerate(ids)}
    random.seed(0)
    idx = OptimizedHNSWIndex(dimension=512)            # defaults: M=16 efC=200 ef=50 (video_search_system.py:79-89)
    t0 = time.time()
    for c0 in range(0, n, 5000):
        idx.add_batch(list(rows[c0:c0 + 5000]), ids[c0:c0 + 5000])
        print(f"  built {min(c0 + 5000, n)} rows, {time.time() - t0:.0f}s", flush=True)
    build_s = time.time() - t0
    stored = np.stack([idx.data[s_] for s_ in ids]).astype(np.float32)
    unit_q = np.stack([q / np.linalg.norm(q) for q in qs]).astype(np.float32)
    ex_rows, ex_dist = knn_oracle.topk(stored, unit_q, 20)     # exact lists by ROW on ties; ties re-ordered by id below
    out = dict(seed=KNN_BIG_SEED, n=n, build_seconds=np.float32(build_s),
               stored_sha256=hashlib.sha256(stored.tobytes()).hexdigest(),
               stored_head=stored[:64].copy(), stored_tail=stored[-64:].copy())
    for ef in (50, n):
        idx.ef_search = ef
        for k in (10, 20):
            t0 = time.time()
            res = [idx.search(q, k) for q in qs]
            dt = time.time() - t0
            assert all(len(rr) == k for rr in res)
            out[f"rows_ef{ef}_k{k}"] = np.array([[row_of[r["id"]] for r in rr] for rr in res], dtype=np.int32)
            out[f"dist_ef{ef}_k{k}"] = np.array([[r["distance"] for r in rr] for rr in res], dtype=np.float32)
            out[f"ms_per_query_ef{ef}_k{k}"] = np.float32(1e3 * dt / len(qs))
            print(f"  ef={ef} k={k}: {1e3 * dt / len(qs):.1f} ms/query", flush=True)
    # the exact answer in the reference's order: (distance, id string)
    exact = []
    for j in range(len(qs)):
        cand = sorted((ex_dist[j, i], ids[ex_rows[j, i]]) for i in range(20))
        exact.append([row_of[s_] for _, s_ in cand])
    exact = np.array(exact, dtype=np.int32)
    out["rows_exact_k20"] = exact
    out["dist_exact_k20"] = ex_dist
    for k in (10, 20):
        got = out[f"rows_ef{n}_k{k}"]
        # a top-k prefix of the exact top-20 is only tie-safe when no tie group straddles rank k; the planted groups sit at ranks 1-3
        out[f"exhaustive_lists_identical_k{k}"] = np.int32(sum(np.array_equal(got[j], exact[j, :k]) for j in range(len(qs))))
        dflt = out[f"rows_ef50_k{k}"]
        out[f"default_ef_recall_k{k}"] = np.float32(np.mean([len(set(dflt[j]) & set(exact[j, :k])) / k for j in range(len(qs))]))
        print(f"  k={k}: exhaustive lists identical to the exact answer for {out[f'exhaustive_lists_identical_k{k}']}/{len(qs)} "
              f"queries; default-ef recall {out[f'default_ef_recall_k{k}']:.4f}", flush=True)
    np.savez_compressed(os.path.join(HERE, f"knn_ref_{n}.npz"), **out)


INTEROP_SEED, INTEROP_N = 11, 50


def interop_vectors():
    """50 seeded rows (dim 64) and 8 queries for the persistence fixtures; ids are the strings the reference's
    caller builds (video_search_system.py:164-166)."""
    rng = np.random.default_rng(INTEROP_SEED)
    vecs = (rng.standard_normal((INTEROP_N, 64)) * 2.5).astype(np.float32)
    ids = [f"video{i // 25}_{i % 25}" for i in range(INTEROP_N)]
    qs = rng.standard_normal((8, 64)).astype(np.float32)
    return vecs, ids, qs


def capture_interop():
    """(i) A file written by the REAL reference HNSWIndex.save (hnsw.py:306-339) + the result lists the real class
    returns from it (ef_search 50 >= N: exhaustive) — the build must load that file and reproduce the lists.
    (ii) The REAL SimpleVideoIndex (video_search_overhaul.py:23-64): the module itself cannot be imported (cv2 is
    not installed: ordinary ImportError), but the class needs only numpy, so its ClassDef is lifted out of the
    reference source with `ast` and executed here; its outputs on seeded rows (un-normalised rows, exact
    duplicates) are the fixture."""
    import ast
    import logging
    import pickle
    from pathlib import Path
    from typing import Any, Dict, List, Optional, Union
    sys.path.insert(0, "/root/reference/src")
    from indexes.hnsw import HNSWIndex                 # the real reference

    vecs, ids, qs = interop_vectors()
    random.seed(3)
    idx = HNSWIndex(dimension=64)
    idx.add_batch(list(vecs), ids)
    path = os.path.join(HERE, "ref_index_50.pkl")
    idx.save(path)
    fresh = HNSWIndex(dimension=64)
    fresh.load(path)
    res = [fresh.search(q, 5) for q in qs]
    np.savez_compressed(os.path.join(HERE, "ref_index_50_results.npz"),
                        ids=np.array([[r["id"] for r in rr] for rr in res]),
                        dist=np.array([[r["distance"] for r in rr] for rr in res], dtype=np.float32),
                        score=np.array([[r["score"] for r in rr] for rr in res], dtype=np.float32),
                        seed=INTEROP_SEED)

    src = open("/root/reference/video_search_overhaul.py").read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "SimpleVideoIndex")
    ns = dict(np=np, pickle=pickle, Path=Path, List=List, Dict=Dict, Any=Any, Optional=Optional, Union=Union,
              logger=logging.getLogger("golden"))
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "video_search_overhaul.py", "exec"), ns)
    SimpleVideoIndex = ns["SimpleVideoIndex"]
    rng = np.random.default_rng(77)
    emb = rng.standard_normal((300, 512)).astype(np.float32)
    emb[:200] /= np.linalg.norm(emb[:200], axis=1, keepdims=True)      # the last 100 rows stay un-normalised
    emb[250] = emb[230]                                                  # exact duplicates -> ties
    emb[20] = emb[10]
    sv = SimpleVideoIndex()
    assert sv.search(emb[0], 3) == []
    for i, e in enumerate(emb):
        sv.add_frame(e, f"video_{i // 100}.mp4", i * 0.5)
    qidx = [0, 7, 10, 123, 230, 260]
    out = {"query_rows": np.array(qidx), "query_scale": np.float32(2.5)}
    for k in (1, 5, 12):
        res = [sv.search(emb[qi] * np.float32(2.5), k) for qi in qidx]
        out[f"frame_id_k{k}"] = np.array([[r["frame_id"] for r in rr] for rr in res], dtype=np.int32)
        out[f"score_k{k}"] = np.array([[r["score"] for r in rr] for rr in res], dtype=np.float64)
        out[f"timestamp_k{k}"] = np.array([[r["timestamp"] for r in rr] for rr in res], dtype=np.float64)
    out["keys"] = np.array(sorted(res[0][0].keys()))
    cache = os.path.join(HERE, "ref_simple_index_cache.pkl")
    small = SimpleVideoIndex()
    for i in range(12):
        small.add_frame(emb[i], f"clip_{i // 6}.mp4", i * 0.25)
    small.video_hashes = {"clip_0.mp4": "abc", "clip_1.mp4": "def"}
    assert small.save_to_disk(Path(cache))
    out["cache_frame_id_k3"] = np.array([r["frame_id"] for r in small.search(emb[5], 3)], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "simple_index.npz"), **out)


if __name__ == "__main__":
    if "knn_big" in sys.argv[1:]:
        capture_knn_big(int(sys.argv[sys.argv.index("knn_big") + 1]))
        sys.exit(0)
    if "interop" in sys.argv[1:]:
        capture_interop()
        sys.exit(0)
    if "l14" in sys.argv[1:]:
        capture_encoder_l14()
        sys.exit(0)
    if "resample" in sys.argv[1:]:
        capture_resample()
        sys.exit(0)
    if "text" in sys.argv[1:]:
        capture_text()
    if "text_l14" in sys.argv[1:]:
        capture_text_l14()
        sys.exit(0)
    emb = capture_encoder()
    capture_knn(emb)
    print("golden vectors written to", HERE)

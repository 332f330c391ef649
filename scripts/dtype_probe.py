"""Which GEMM groups can stay bf16 under the 1e-3 score tolerance?  Encodes the 64 golden frames with every
operand-type assignment of interest and prints the error against the transformers fp32 golden embeddings.
GPU box only: python scripts/dtype_probe.py"""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import INDEX_SEED, synth_frames
from video_quierer_amd.encoder import VitEncoder, DTYPE_GROUPS
from video_quierer_amd.weights import VIT_B_32, seeded_weights

gold = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "encoder_b32_seed1234.npz"))["embeddings"]
rows = np.random.default_rng(INDEX_SEED).standard_normal((1000, 512)).astype(np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
w = seeded_weights(VIT_B_32, 1234)
frames = synth_frames(64)
names = list(DTYPE_GROUPS)
combos = [()] + [(n,) for n in names] + [("fc1", "fc2"), ("qkv", "attn"), ("qkv", "fc1"), ("qkv", "fc1", "fc2"),
                                         ("qkv", "attn", "fc1", "fc2"), ("patch", "qkv", "fc1", "fc2"), tuple(names)]
for c in combos:
    enc = VitEncoder(VIT_B_32, w, max_batch=64, compute_dtype="fp16:" + "+".join(c) if c else "bf16")
    emb = enc.encode(frames)
    enc.close()
    l2 = np.linalg.norm(emb - gold, axis=1)
    diff = np.abs(emb @ rows.T - gold @ rows.T)
    self_sc = np.abs(emb @ emb.T - gold @ gold.T)
    print(f"fp16 groups {'+'.join(c) or '(none)':28s} L2 err max {l2.max():.2e} rms {np.sqrt((l2**2).mean()):.2e} | "
          f"score diff vs 1000 rows max {diff.max():.2e} rms {np.sqrt((diff**2).mean()):.2e} | "
          f"frame-frame score diff max {self_sc.max():.2e}", flush=True)

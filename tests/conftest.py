import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
FRAME_SEED = 20250824
WEIGHT_SEED = 1234
INDEX_SEED = 7


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def synth_frames(n, seed=FRAME_SEED):
    """The reference's own synthetic-frame convention (src/video_search_system.py:556-557)."""
    return np.random.default_rng(seed).integers(0, 255, (n, 224, 224, 3), dtype=np.uint8)


@pytest.fixture(scope="session")
def golden_encoder():
    return np.load(os.path.join(GOLDEN, "encoder_b32_seed1234.npz"))


@pytest.fixture(scope="session")
def golden_knn():
    return np.load(os.path.join(GOLDEN, "knn_cfg1.npz"))


@pytest.fixture(scope="session")
def b32_weights():
    from video_quierer_amd.weights import VIT_B_32, seeded_weights
    return seeded_weights(VIT_B_32, WEIGHT_SEED)


def outlier_weights(seed=WEIGHT_SEED):
    """Seeded ViT-B/32 weights reshaped towards what trained CLIP checkpoints look like (parity on
    N(0, 1/fan_in) weights alone says nothing about outlier channels): three "massive activation" channels that two
    MLP blocks write into the residual stream (|x| of 25-60 beside O(1) neighbours), LayerNorm gains spread over a
    log-normal with the hot channels damped in the later blocks, a large class token, a pre-LayerNorm that amplifies
    the hot channels, and heavy-tailed linear weights (0.5 % of the entries six times larger)."""
    from video_quierer_amd.weights import VIT_B_32, seeded_weights
    W = {k: v.copy() for k, v in seeded_weights(VIT_B_32, seed).items()}
    rng = np.random.default_rng([seed, 77])
    hot = np.array([7, 133, 520])
    for name, w in W.items():
        if name.endswith(("q_proj.weight", "k_proj.weight", "v_proj.weight", "out_proj.weight", "fc1.weight", "fc2.weight")):
            mask = rng.random(w.shape) < 0.005
            w[mask] *= 6.0
        elif name.endswith(("layer_norm1.weight", "layer_norm2.weight")):
            w *= np.exp(0.5 * rng.standard_normal(w.shape)).astype(np.float32)
    for l, amp in ((2, (40.0, -60.0, 25.0)), (5, (-15.0, 30.0, 20.0))):
        W[f"vision_model.encoder.layers.{l}.mlp.fc2.bias"][hot] += np.array(amp, np.float32)
        W[f"vision_model.encoder.layers.{l}.mlp.fc2.weight"][hot, :] *= 4.0
    for l in range(3, 12):
        for ln in ("layer_norm1", "layer_norm2"):
            W[f"vision_model.encoder.layers.{l}.{ln}.weight"][hot] *= 0.1
    W["vision_model.pre_layrnorm.weight"][hot] *= 6.0
    W["vision_model.embeddings.class_embedding"] *= 5.0
    return W


@pytest.fixture(scope="session")
def gpu_lib():
    """Binds the GPU once; GPU tests fail (not skip) if the native library cannot run."""
    from video_quierer_amd import _lib
    _lib.init(0)
    return _lib


# ---- the configs[2] pin against the REAL reference above N = 1,000 (tests/golden/knn_ref_<n>.npz) ----
KNN_BIG_SEED = 20251005
KNN_BIG_DUPES = ((2, 10), (3, 20, 100), (777, 5000))      # groups of rows made exact duplicates of the first one


def knn_big_ids(n):
    """The ids the reference's caller builds: f"{video_id}_{i}" (src/video_search_system.py:164-166), four videos."""
    per = n // 4
    return [f"video{r // per}_{r % per}" for r in range(n)]


def knn_big_inputs(n, nq=64):
    """SURVEY.md §8(d) config 3's recipe at prefix size: `standard_normal` fp32 rows (normalised by `add`), queries from a
    different seed.  Planted: groups of exact duplicate rows whose string ids sort differently from their row numbers
    ("video0_10" < "video0_2"), and queries next to them, so the (distance, id) tie rule of hnsw.py:269/518 is exercised
    with the caller's ids against the real class."""
    rows = np.random.default_rng(KNN_BIG_SEED).standard_normal((n, 512)).astype(np.float32)
    for grp in KNN_BIG_DUPES:
        for r in grp[1:]:
            if r < n and grp[0] < n:
                rows[r] = rows[grp[0]]
    qrng = np.random.default_rng(KNN_BIG_SEED + 1)
    qs = qrng.standard_normal((nq, 512)).astype(np.float32)
    for j, grp in enumerate(KNN_BIG_DUPES):
        if grp[0] < n:
            qs[j] = rows[grp[0]] + np.float32(0.05) * qrng.standard_normal(512).astype(np.float32)
    return rows, qs


# ---- resize fixtures (tests/golden/resample_pil.npz; inputs are regenerated from the seed) ----
RESAMPLE_SEED = 31337
RESAMPLE_CASES = [            # (name, h, w, kind, mode)   kind: noise | smooth ; mode: stretch | clip
    ("noise_97x131_stretch", 97, 131, "noise", "stretch"),
    ("noise_300x400_clip", 300, 400, "noise", "clip"),
    ("smooth_1080x1920_stretch", 1080, 1920, "smooth", "stretch"),
    ("noise_1080x1920_clip", 1080, 1920, "noise", "clip"),
    ("noise_719x405_clip", 719, 405, "noise", "clip"),
    ("noise_224x224_stretch", 224, 224, "noise", "stretch"),
    ("noise_100x80_stretch", 100, 80, "noise", "stretch"),          # upscale
    ("noise_225x1000_clip", 225, 1000, "noise", "clip"),
]


def resample_input(h, w, kind, seed=RESAMPLE_SEED):
    """Seeded test frames: uniform noise, or a smooth ramp + low-amplitude noise (what video frames look like)."""
    rng = np.random.default_rng([seed, h, w])
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([yy * 255 // max(h - 1, 1), xx * 255 // max(w - 1, 1), (yy + xx) % 256], -1)
    return np.clip(base + rng.integers(-6, 7, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.fixture(scope="session")
def golden_resample():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resample_pil.npz"))


# ---- a local HF-style checkpoint directory (the only non-seeded way in: feature_extractor.py:76-77) ----
TOY_MERGES = [("c", "a"), ("ca", "t</w>"), ("d", "o"), ("do", "g</w>"), ("t", "h"), ("th", "e</w>")]


def toy_text_config():
    """Text tower sized for the toy tokenizer below (520 tokens: 256 byte symbols, their end-of-word forms, six
    merges, <|startoftext|> = 518, <|endoftext|> = 519); everything else is CLIP ViT-B/32's text geometry."""
    from video_quierer_amd.weights import TextConfig
    return TextConfig(vocab=520, eos_token_id=519, bos_token_id=518)


def write_checkpoint_dir(path, seed=WEIGHT_SEED, dtype="float32"):
    """model.safetensors (both towers, HF state_dict names, plus the extra tensors a real CLIP checkpoint carries) +
    config.json + vocab.json / merges.txt of a byte-level BPE the transformers CLIPTokenizer accepts."""
    import json
    from video_quierer_amd.weights import VIT_B_32, seeded_text_weights, seeded_weights
    os.makedirs(path, exist_ok=True)
    tensors = dict(seeded_weights(VIT_B_32, seed))
    tensors.update(seeded_text_weights(toy_text_config(), seed))
    tensors["logit_scale"] = np.array(4.6052, dtype=np.float32)                        # ignored by the loader
    tensors["vision_model.embeddings.position_ids"] = np.arange(50, dtype=np.int64)[None]
    if dtype == "float32":
        from safetensors.numpy import save_file
        save_file(tensors, os.path.join(path, "model.safetensors"))
    else:
        import torch
        from safetensors.torch import save_file
        td = getattr(torch, dtype)
        save_file({k: (torch.from_numpy(v).to(td) if v.dtype == np.float32 else torch.from_numpy(v)) for k, v in tensors.items()},
                  os.path.join(path, "model.safetensors"))
    cfg = {"model_type": "clip", "projection_dim": 512,
           "vision_config": {"image_size": 224, "patch_size": 32, "hidden_size": 768, "intermediate_size": 3072,
                             "num_hidden_layers": 12, "num_attention_heads": 12, "layer_norm_eps": 1e-5},
           "text_config": {"vocab_size": 520, "max_position_embeddings": 77, "hidden_size": 512, "intermediate_size": 2048,
                           "num_hidden_layers": 12, "num_attention_heads": 8, "eos_token_id": 519, "bos_token_id": 518,
                           "layer_norm_eps": 1e-5}}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    chars = [chr(c) for c in cs]
    vocab = chars + [c + "</w>" for c in chars] + [a + b for a, b in TOY_MERGES] + ["<|startoftext|>", "<|endoftext|>"]
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump({t: i for i, t in enumerate(vocab)}, f)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in TOY_MERGES) + "\n")
    return path

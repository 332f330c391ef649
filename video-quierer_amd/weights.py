"""CLIP vision-tower weights: canonical tensor order, deterministic seeded
generation and loading from a local HF checkpoint directory.

The reference obtains weights with ``CLIPModel.from_pretrained(model_name)``
(reference src/core/feature_extractor.py:76-77), a by-name hub fetch that is
unavailable offline.  ``from_pretrained`` also accepts a local directory; that
is what :func:`load_weights` supports (``*.safetensors`` with the HF key
names).  With no checkpoint available the spec ``"seed:<int>"`` generates a
deterministic weight set of the same shapes (throughput is weight-value
independent; parity is checked against the fp32 oracle on the same weights).

The order of :func:`weight_names` is the order of the ``weights`` pointer
array handed to ``vq_encoder_create`` (include/vq_amd.h).
"""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

import numpy as np


@dataclass(frozen=True)
class VitConfig:
    """Vision-tower geometry (transformers models/clip/configuration_clip.py:97-106,160)."""

    image_size: int = 224
    patch_size: int = 32
    hidden: int = 768
    mlp: int = 3072
    layers: int = 12
    heads: int = 12
    proj_dim: int = 512
    ln_eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def patches(self) -> int:
        return self.grid * self.grid

    @property
    def tokens(self) -> int:
        return self.patches + 1

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def patch_k(self) -> int:
        return 3 * self.patch_size * self.patch_size

    def macs_per_frame(self) -> int:
        """Multiply-accumulates of one full forward pass."""
        t, h, m = self.tokens, self.hidden, self.mlp
        layer = t * (4 * h * h + 2 * h * m) + 2 * self.heads * t * t * self.head_dim
        return self.patches * self.patch_k * h + self.layers * layer + h * self.proj_dim

    def as_dict(self) -> dict:
        return asdict(self)


VIT_B_32 = VitConfig()
VIT_L_14_336 = VitConfig(image_size=336, patch_size=14, hidden=1024, mlp=4096,
                         layers=24, heads=16, proj_dim=768)

_KNOWN = {
    "openai/clip-vit-base-patch32": VIT_B_32,
    "clip-vit-base-patch32": VIT_B_32,
    "openai/clip-vit-large-patch14-336": VIT_L_14_336,
    "clip-vit-large-patch14-336": VIT_L_14_336,
}


def weight_shapes(cfg: VitConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Canonical (HF state_dict name, shape) list, in C-ABI order."""
    h, m, p = cfg.hidden, cfg.mlp, cfg.patch_size
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("vision_model.embeddings.class_embedding", (h,)),
        ("vision_model.embeddings.patch_embedding.weight", (h, 3, p, p)),
        ("vision_model.embeddings.position_embedding.weight", (cfg.tokens, h)),
        ("vision_model.pre_layrnorm.weight", (h,)),   # sic: HF key spelling
        ("vision_model.pre_layrnorm.bias", (h,)),
    ]
    for l in range(cfg.layers):
        pre = f"vision_model.encoder.layers.{l}."
        out += [
            (pre + "layer_norm1.weight", (h,)),
            (pre + "layer_norm1.bias", (h,)),
            (pre + "self_attn.q_proj.weight", (h, h)),
            (pre + "self_attn.q_proj.bias", (h,)),
            (pre + "self_attn.k_proj.weight", (h, h)),
            (pre + "self_attn.k_proj.bias", (h,)),
            (pre + "self_attn.v_proj.weight", (h, h)),
            (pre + "self_attn.v_proj.bias", (h,)),
            (pre + "self_attn.out_proj.weight", (h, h)),
            (pre + "self_attn.out_proj.bias", (h,)),
            (pre + "layer_norm2.weight", (h,)),
            (pre + "layer_norm2.bias", (h,)),
            (pre + "mlp.fc1.weight", (m, h)),
            (pre + "mlp.fc1.bias", (m,)),
            (pre + "mlp.fc2.weight", (h, m)),
            (pre + "mlp.fc2.bias", (h,)),
        ]
    out += [
        ("vision_model.post_layernorm.weight", (h,)),
        ("vision_model.post_layernorm.bias", (h,)),
        ("visual_projection.weight", (cfg.proj_dim, h)),
    ]
    return out


def weight_names(cfg: VitConfig) -> List[str]:
    return [n for n, _ in weight_shapes(cfg)]


def seeded_weights(cfg: VitConfig, seed: int) -> Dict[str, np.ndarray]:
    """Deterministic fp32 weights: tensor i is drawn from PCG64([seed, i]).

    Scales are chosen so activations stay O(1) through the stack (linear
    weights ~ N(0, 1/fan_in), attention logits ~ unit variance), which makes
    the softmax and both LayerNorm parameters matter in the parity check.
    """
    out: Dict[str, np.ndarray] = {}
    for i, (name, shape) in enumerate(weight_shapes(cfg)):
        rng = np.random.Generator(np.random.PCG64([int(seed), i]))
        z = rng.standard_normal(shape, dtype=np.float32)
        if name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight"):
            w = 1.0 + 0.1 * z
        elif name.endswith("norm.bias") or name.endswith("norm1.bias") or name.endswith("norm2.bias"):
            w = 0.05 * z
        elif name.endswith(".bias"):
            w = 0.02 * z
        elif name.endswith("class_embedding"):
            w = z * (cfg.hidden ** -0.5) * 4.0
        elif name.endswith("position_embedding.weight"):
            w = 0.1 * z
        elif name.endswith("patch_embedding.weight"):
            w = z * (cfg.patch_k ** -0.5)
        else:  # nn.Linear [out, in]
            w = z * (shape[1] ** -0.5)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


# ---- text tower (CLIPTextModel + text_projection; transformers modeling_clip.py:222-256, 500-586) ----
@dataclass(frozen=True)
class TextConfig:
    vocab: int = 49408
    max_positions: int = 77
    hidden: int = 512
    mlp: int = 2048
    layers: int = 12
    heads: int = 8
    proj_dim: int = 512
    eos_token_id: int = 49407
    bos_token_id: int = 49406
    ln_eps: float = 1e-5


TEXT_B_32 = TextConfig()
TEXT_L_14 = TextConfig(hidden=768, mlp=3072, heads=12, proj_dim=768)


def text_weight_shapes(cfg: TextConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Canonical (HF state_dict name, shape) list of the text tower, in C-ABI order."""
    h, m = cfg.hidden, cfg.mlp
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("text_model.embeddings.token_embedding.weight", (cfg.vocab, h)),
        ("text_model.embeddings.position_embedding.weight", (cfg.max_positions, h)),
    ]
    for l in range(cfg.layers):
        pre = f"text_model.encoder.layers.{l}."
        out += [
            (pre + "layer_norm1.weight", (h,)), (pre + "layer_norm1.bias", (h,)),
            (pre + "self_attn.q_proj.weight", (h, h)), (pre + "self_attn.q_proj.bias", (h,)),
            (pre + "self_attn.k_proj.weight", (h, h)), (pre + "self_attn.k_proj.bias", (h,)),
            (pre + "self_attn.v_proj.weight", (h, h)), (pre + "self_attn.v_proj.bias", (h,)),
            (pre + "self_attn.out_proj.weight", (h, h)), (pre + "self_attn.out_proj.bias", (h,)),
            (pre + "layer_norm2.weight", (h,)), (pre + "layer_norm2.bias", (h,)),
            (pre + "mlp.fc1.weight", (m, h)), (pre + "mlp.fc1.bias", (m,)),
            (pre + "mlp.fc2.weight", (h, m)), (pre + "mlp.fc2.bias", (h,)),
        ]
    out += [
        ("text_model.final_layer_norm.weight", (h,)),
        ("text_model.final_layer_norm.bias", (h,)),
        ("text_projection.weight", (cfg.proj_dim, h)),
    ]
    return out


def seeded_text_weights(cfg: TextConfig, seed: int) -> Dict[str, np.ndarray]:
    """Deterministic fp32 text-tower weights (tensor i from PCG64([seed, 1000 + i]))."""
    out: Dict[str, np.ndarray] = {}
    for i, (name, shape) in enumerate(text_weight_shapes(cfg)):
        rng = np.random.Generator(np.random.PCG64([int(seed), 1000 + i]))
        z = rng.standard_normal(shape, dtype=np.float32)
        if "layer_norm" in name and name.endswith(".weight"):
            w = 1.0 + 0.1 * z
        elif "layer_norm" in name and name.endswith(".bias"):
            w = 0.05 * z
        elif name.endswith(".bias"):
            w = 0.02 * z
        elif "token_embedding" in name:
            w = 0.5 * z
        elif "position_embedding" in name:
            w = 0.1 * z
        else:
            w = z * (shape[1] ** -0.5)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def resolve_text_model(model_name: str) -> Tuple[TextConfig, Dict[str, np.ndarray], str]:
    """``model_name`` -> (text config, weights, tokenizer directory or "").  Same rules as resolve_model."""
    if model_name.startswith("seed:"):
        parts = model_name.split(":")
        arch = parts[2] if len(parts) > 2 else "b32"
        cfg = {"b32": TEXT_B_32, "l14-336": TEXT_L_14}[arch]
        return cfg, seeded_text_weights(cfg, int(parts[1])), ""
    path = model_name
    if not os.path.isdir(path):
        root = os.environ.get("VQ_AMD_MODEL_DIR")
        cand = os.path.join(root, os.path.basename(model_name)) if root else None
        if cand and os.path.isdir(cand):
            path = cand
        else:
            raise FileNotFoundError(f"model {model_name!r}: no local checkpoint directory (this build never downloads)")
    cfg = TEXT_B_32
    cj = os.path.join(path, "config.json")
    if os.path.exists(cj):
        import json
        with open(cj) as f:
            full = json.load(f)
        t = full.get("text_config", {})
        cfg = TextConfig(vocab=t.get("vocab_size", 49408), max_positions=t.get("max_position_embeddings", 77),
                         hidden=t.get("hidden_size", 512), mlp=t.get("intermediate_size", 2048),
                         layers=t.get("num_hidden_layers", 12), heads=t.get("num_attention_heads", 8),
                         proj_dim=full.get("projection_dim", 512), eos_token_id=t.get("eos_token_id", 49407),
                         bos_token_id=t.get("bos_token_id", 49406), ln_eps=t.get("layer_norm_eps", 1e-5))
    want = dict(text_weight_shapes(cfg))
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path!r}")
    from safetensors import safe_open
    got: Dict[str, np.ndarray] = {}
    for fn in files:
        with safe_open(fn, framework="pt") as f:
            for key in f.keys():
                if key in want:
                    got[key] = np.ascontiguousarray(f.get_tensor(key).float().numpy())
    missing = [k for k in want if k not in got]
    if missing:
        raise KeyError(f"checkpoint {path!r} lacks {len(missing)} text tensors, e.g. {missing[:3]}")
    return cfg, got, path


def _load_dir(path: str, cfg: VitConfig) -> Dict[str, np.ndarray]:
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path!r}")
    from safetensors import safe_open  # local import: only needed for real checkpoints

    want = dict(weight_shapes(cfg))
    got: Dict[str, np.ndarray] = {}
    for fn in files:
        with safe_open(fn, framework="pt") as f:  # pt: handles bf16/fp16 checkpoints
            for key in f.keys():
                if key in want:
                    got[key] = np.ascontiguousarray(f.get_tensor(key).float().numpy())
    missing = [k for k in want if k not in got]
    if missing:
        raise KeyError(f"checkpoint {path!r} lacks {len(missing)} tensors, e.g. {missing[:3]}")
    for k, shp in want.items():
        if tuple(got[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {got[k].shape} != expected {shp}")
    return got


def resolve_model(model_name: str) -> Tuple[VitConfig, Dict[str, np.ndarray]]:
    """``model_name`` → (config, weights).

    * ``"seed:<int>"`` or ``"seed:<int>:<arch>"`` — seeded weights
      (arch ``b32`` default, ``l14-336``).
    * a local directory holding HF safetensors (+ optional config.json).
    * a known hub name: only resolves if ``$VQ_AMD_MODEL_DIR/<basename>`` exists;
      this build never fetches.
    """
    if model_name.startswith("seed:"):
        parts = model_name.split(":")
        seed = int(parts[1])
        arch = parts[2] if len(parts) > 2 else "b32"
        cfg = {"b32": VIT_B_32, "l14-336": VIT_L_14_336}[arch]
        return cfg, seeded_weights(cfg, seed)
    path = model_name
    if not os.path.isdir(path):
        root = os.environ.get("VQ_AMD_MODEL_DIR")
        cand = os.path.join(root, os.path.basename(model_name)) if root else None
        if cand and os.path.isdir(cand):
            path = cand
        else:
            raise FileNotFoundError(
                f"model {model_name!r}: not a local directory and no copy under $VQ_AMD_MODEL_DIR; "
                "this build never downloads weights — pass a checkpoint directory or 'seed:<int>'")
    cfg = _KNOWN.get(model_name) or _KNOWN.get(os.path.basename(os.path.normpath(path)))
    cj = os.path.join(path, "config.json")
    if os.path.exists(cj):
        import json
        with open(cj) as f:
            v = json.load(f).get("vision_config", {})
        if v:
            cfg = VitConfig(image_size=v.get("image_size", 224), patch_size=v.get("patch_size", 32),
                            hidden=v.get("hidden_size", 768), mlp=v.get("intermediate_size", 3072),
                            layers=v.get("num_hidden_layers", 12), heads=v.get("num_attention_heads", 12),
                            proj_dim=json.load(open(cj)).get("projection_dim", 512),
                            ln_eps=v.get("layer_norm_eps", 1e-5))
    if cfg is None:
        cfg = VIT_B_32
    return cfg, _load_dir(path, cfg)

"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU fp32 restatement of the reference's frame-embedding path:
``FeatureExtractor.extract_batch`` (reference src/core/feature_extractor.py:137-177)
with the CLIP vision tower it calls from third-party ``transformers``
(unpinned in the reference's requirements.txt:4; 5.15.0 installed where the
golden vectors were captured; "tf:" below = transformers/models/clip/modeling_clip.py).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  It is pinned against the real
``transformers.CLIPModel`` by tests/golden/encoder_b32_seed1234.npz (captured by
tests/golden/make_golden.py in the build container; see that script).

Plain torch-CPU tensor ops only (matmul / softmax / mean / var): no nn.Module,
no transformers import, nothing from the product package.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch

# reference src/core/feature_extractor.py:57-60 (torchvision Normalize constants)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_u8(frames: np.ndarray, swap_rb: bool = True) -> torch.Tensor:
    """uint8 [n,H,W,3] → fp32 [n,3,H,W].

    Follows ``_preprocess_image`` (feature_extractor.py:105-116): an ndarray
    input is treated as BGR and channel-reversed (cv2.COLOR_BGR2RGB, :111-112);
    ``Resize((224,224))`` is the identity at 224×224 (what frame_extractor.py
    :283-284 guarantees upstream); ``ToTensor`` = HWC→CHW and /255;
    ``Normalize`` = (x-mean)/std per channel (:54-61).
    """
    x = np.asarray(frames)
    assert x.dtype == np.uint8 and x.ndim == 4 and x.shape[-1] == 3
    if swap_rb:
        x = x[..., ::-1]
    t = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2).to(torch.float32) / 255.0
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(1, 3, 1, 1)
    return (t - mean) / std


def _ln(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    # nn.LayerNorm: biased variance over the last axis
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def vit_forward(pixel: torch.Tensor, W: Dict[str, torch.Tensor], *, patch: int, heads: int,
                layers: int, eps: float = 1e-5, return_hidden: bool = False) -> torch.Tensor:
    """fp32 [n,3,H,W] → fp32 [n,proj_dim] (un-normalised image features).

    tf:202-218 embeddings, :641-642 pre-LN, :362-383 encoder layer (pre-LN
    residual block), :298-335 + :259-277 attention (fp32 softmax, scale
    d_h**-0.5, no mask), :346-350 MLP with quick_gelu (tf activations.py:117-123:
    x*sigmoid(1.702x)), :649-651 CLS pooling + post-LN, :719-753 visual_projection
    (no bias).
    """
    n, c, H, Wd = pixel.shape
    g_h, g_w = H // patch, Wd // patch
    hid = W["vision_model.embeddings.class_embedding"].shape[0]
    # Conv2d(k=stride=patch, no bias) as a GEMM over (c, ky, kx)-ordered patches
    p = pixel.reshape(n, c, g_h, patch, g_w, patch).permute(0, 2, 4, 1, 3, 5)
    p = p.reshape(n, g_h * g_w, c * patch * patch)
    wp = W["vision_model.embeddings.patch_embedding.weight"].reshape(hid, -1)
    x = p @ wp.t()                                                   # [n,P,hid]
    cls = W["vision_model.embeddings.class_embedding"].expand(n, 1, hid)
    x = torch.cat([cls, x], dim=1) + W["vision_model.embeddings.position_embedding.weight"]
    x = _ln(x, W["vision_model.pre_layrnorm.weight"], W["vision_model.pre_layrnorm.bias"], eps)
    T = x.shape[1]
    dh = hid // heads
    scale = dh ** -0.5
    for l in range(layers):
        pre = f"vision_model.encoder.layers.{l}."
        h = _ln(x, W[pre + "layer_norm1.weight"], W[pre + "layer_norm1.bias"], eps)
        q = h @ W[pre + "self_attn.q_proj.weight"].t() + W[pre + "self_attn.q_proj.bias"]
        k = h @ W[pre + "self_attn.k_proj.weight"].t() + W[pre + "self_attn.k_proj.bias"]
        v = h @ W[pre + "self_attn.v_proj.weight"].t() + W[pre + "self_attn.v_proj.bias"]
        q = q.view(n, T, heads, dh).transpose(1, 2)
        k = k.view(n, T, heads, dh).transpose(1, 2)
        v = v.view(n, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * scale
        a = torch.softmax(s, dim=-1, dtype=torch.float32)
        o = (a @ v).transpose(1, 2).reshape(n, T, hid)
        x = x + (o @ W[pre + "self_attn.out_proj.weight"].t() + W[pre + "self_attn.out_proj.bias"])
        h = _ln(x, W[pre + "layer_norm2.weight"], W[pre + "layer_norm2.bias"], eps)
        h = h @ W[pre + "mlp.fc1.weight"].t() + W[pre + "mlp.fc1.bias"]
        h = h * torch.sigmoid(1.702 * h)
        x = x + (h @ W[pre + "mlp.fc2.weight"].t() + W[pre + "mlp.fc2.bias"])
    if return_hidden:            # residual stream after `layers` blocks (layer-wise parity tests)
        return x
    pooled = _ln(x[:, 0, :], W["vision_model.post_layernorm.weight"],
                 W["vision_model.post_layernorm.bias"], eps)
    return pooled @ W["visual_projection.weight"].t()


def text_forward(input_ids: torch.Tensor, W: Dict[str, torch.Tensor], *, heads: int, layers: int,
                 eos_token_id: int, eps: float = 1e-5) -> torch.Tensor:
    """int64 [n,L] token ids -> fp32 [n,proj_dim] (un-normalised text features).

    Restates what ``extract_text_features`` (reference src/core/feature_extractor.py:218-234) gets from
    ``CLIPModel.get_text_features``: tf:222-256 token + position embeddings, tf:543-554 causal mask,
    the same pre-LN blocks as the vision tower (tf:362-383), tf:566 final_layer_norm, tf:568-586 pooling at
    the first EOS position (argmax of the ids for the legacy eos_token_id == 2), text_projection (no bias).
    """
    n, L = input_ids.shape
    p = "text_model."
    x = W[p + "embeddings.token_embedding.weight"][input_ids] + W[p + "embeddings.position_embedding.weight"][:L]
    hid = x.shape[-1]
    dh = hid // heads
    mask = torch.full((L, L), float("-inf")).triu(1)
    for l in range(layers):
        pre = f"{p}encoder.layers.{l}."
        h = _ln(x, W[pre + "layer_norm1.weight"], W[pre + "layer_norm1.bias"], eps)
        q = (h @ W[pre + "self_attn.q_proj.weight"].t() + W[pre + "self_attn.q_proj.bias"]).view(n, L, heads, dh).transpose(1, 2)
        k = (h @ W[pre + "self_attn.k_proj.weight"].t() + W[pre + "self_attn.k_proj.bias"]).view(n, L, heads, dh).transpose(1, 2)
        v = (h @ W[pre + "self_attn.v_proj.weight"].t() + W[pre + "self_attn.v_proj.bias"]).view(n, L, heads, dh).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5 + mask, dim=-1, dtype=torch.float32)
        o = (a @ v).transpose(1, 2).reshape(n, L, hid)
        x = x + (o @ W[pre + "self_attn.out_proj.weight"].t() + W[pre + "self_attn.out_proj.bias"])
        h = _ln(x, W[pre + "layer_norm2.weight"], W[pre + "layer_norm2.bias"], eps)
        h = h @ W[pre + "mlp.fc1.weight"].t() + W[pre + "mlp.fc1.bias"]
        h = h * torch.sigmoid(1.702 * h)
        x = x + (h @ W[pre + "mlp.fc2.weight"].t() + W[pre + "mlp.fc2.bias"])
    x = _ln(x, W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"], eps)
    if eos_token_id == 2:
        pos = input_ids.argmax(dim=-1)
    else:
        pos = (input_ids == eos_token_id).int().argmax(dim=-1)
    pooled = x[torch.arange(n), pos]
    return pooled @ W["text_projection.weight"].t()


def encode_token_ids(input_ids: np.ndarray, weights: Dict[str, np.ndarray], *, heads: int = 8, layers: int = 12,
                     eos_token_id: int = 49407, eps: float = 1e-5) -> np.ndarray:
    """int [n,L] -> np.float32 [n,proj_dim], L2-normalised (feature_extractor.py:226-230)."""
    W = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in weights.items()}
    with torch.no_grad():
        f = text_forward(torch.from_numpy(np.asarray(input_ids, dtype=np.int64)), W, heads=heads, layers=layers,
                         eos_token_id=eos_token_id, eps=eps)
        return l2_normalize(f).numpy().astype(np.float32)


def l2_normalize(f: torch.Tensor) -> torch.Tensor:
    """F.normalize(p=2, dim=1): x / max(||x||, 1e-12) (feature_extractor.py:157)."""
    return f / f.norm(dim=1, keepdim=True).clamp_min(1e-12)


def encode_frames(frames: np.ndarray, weights: Dict[str, np.ndarray], *, patch: int = 32,
                  heads: int = 12, layers: int = 12, eps: float = 1e-5, swap_rb: bool = True,
                  batch_size: int = 32) -> np.ndarray:
    """uint8 [n,H,W,3] → np.float32 [n,proj_dim], L2-normalised.

    Mirrors extract_batch (feature_extractor.py:137-160) applied in
    ``batch_size`` slices as extract_from_video_frames does (:191-193).
    """
    if len(frames) == 0:
        return np.array([])            # feature_extractor.py:142-143
    W = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in weights.items()}
    outs = []
    with torch.no_grad():
        for i in range(0, len(frames), batch_size):
            px = preprocess_u8(np.asarray(frames[i:i + batch_size]), swap_rb)
            f = vit_forward(px, W, patch=patch, heads=heads, layers=layers, eps=eps)
            outs.append(l2_normalize(f).numpy().astype(np.float32))
    return np.concatenate(outs, axis=0)


def extract_from_video_frames(frames_data: Sequence[dict], weights, *, batch_size: int = 32, **kw):
    """Restates extract_from_video_frames (feature_extractor.py:179-209): copies of
    the input dicts, in order, plus 'features' and 'feature_extraction_time'."""
    import time
    if not frames_data:
        return []
    t0 = time.time()
    out = []
    for i in range(0, len(frames_data), batch_size):
        chunk = frames_data[i:i + batch_size]
        feats = encode_frames(np.stack([d["frame"] for d in chunk]), weights,
                              batch_size=batch_size, **kw)
        for d, f in zip(chunk, feats):
            r = dict(d)
            r["features"] = f
            r["feature_extraction_time"] = time.time() - t0
            out.append(r)
    return out

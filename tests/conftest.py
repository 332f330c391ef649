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


@pytest.fixture(scope="session")
def gpu_lib():
    """Binds the GPU once; GPU tests fail (not skip) if the native library cannot run."""
    from video_quierer_amd import _lib
    _lib.init(0)
    return _lib


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

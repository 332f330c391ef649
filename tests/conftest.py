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

import os
import sys

import numpy as np
import pytest

try:  # before anything loads libmrgfe: a process that also uses PyTorch must bind both to ONE HIP runtime, torch's (see _lib.lib)
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


def small_cloud(n=2000, seed=0, extent=(20.0, 12.0, 3.0)):
    """Structured random cloud: a ground plane, two walls and scattered clutter (N x 4 float32)."""
    rng = np.random.default_rng(seed)
    n_g, n_w = int(n * 0.45), int(n * 0.2)
    n_c = n - n_g - 2 * n_w
    g = np.stack([rng.uniform(-extent[0], extent[0], n_g), rng.uniform(-extent[1], extent[1], n_g), -1.73 + rng.normal(0, 0.02, n_g)], 1)
    w1 = np.stack([rng.uniform(-extent[0], extent[0], n_w), extent[1] * 0.9637 + rng.normal(0, 0.02, n_w), rng.uniform(-1.7, extent[2], n_w)], 1)
    w2 = np.stack([extent[0] * 0.5817 + rng.normal(0, 0.02, n_w), rng.uniform(-extent[1], extent[1], n_w), rng.uniform(-1.7, extent[2], n_w)], 1)
    c = np.stack([rng.uniform(-extent[0], extent[0], n_c), rng.uniform(-extent[1], extent[1], n_c), rng.uniform(-1.7, extent[2], n_c)], 1)
    xyz = np.concatenate([g, w1, w2, c]).astype(np.float32)
    xyz = xyz[rng.permutation(len(xyz))]
    out = np.empty((len(xyz), 4), dtype=np.float32)
    out[:, :3] = xyz
    out[:, 3] = rng.uniform(0, 1, len(xyz)).astype(np.float32)
    return out


@pytest.fixture(scope="session")
def street_pair_vlp16():
    """VLP-16 pair of the synthetic street (BASELINE config 1 shape: ~15-25k points per scan)."""
    from mrg_slam_amd import synth

    sc = synth.street_scene()
    return synth.scan_pair(0, "VLP16", sc)

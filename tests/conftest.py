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


# tests/faultinject/ needs the -DMRGFE_TESTING build of the library: collected only in the child process tests/test_gpu_hardening.py starts for it
collect_ignore = [] if os.environ.get("MRGFE_FAULTINJECT") else ["faultinject"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


from oracle.replay import small_cloud  # noqa: E402,F401  (one definition: the soak of oracle/replay.py and bench.py use it too)


@pytest.fixture(scope="session")
def street_pair_vlp16():
    """VLP-16 pair of the synthetic street (BASELINE config 1 shape: ~15-25k points per scan)."""
    from mrg_slam_amd import synth

    sc = synth.street_scene()
    return synth.scan_pair(0, "VLP16", sc)

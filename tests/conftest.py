import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def small_seq(oracle):
    """6 synthetic HDL-64 scans at reduced azimuth resolution (500 steps)."""
    w = oracle.S1World(n_az=500)
    poses = w.trajectory(6)
    xyzi, off = w.scans(poses)
    return dict(world=w, poses=poses, xyzi=xyzi, off=off)


@pytest.fixture(scope="session")
def full_seq(oracle):
    """3 synthetic HDL-64 scans at the full 2000 azimuth steps (~118 k points each)."""
    w = oracle.S1World()
    poses = w.trajectory(3)
    xyzi, off = w.scans(poses)
    return dict(world=w, poses=poses, xyzi=xyzi, off=off)


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    import lmono_amd
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    return lmono_amd.Context(0)

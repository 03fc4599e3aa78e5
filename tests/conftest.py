import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy reader of a committed fixture (tests/golden/<name>.npz) returning torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __contains__(self, key):
        return key in self._z.files

    def __getitem__(self, key):
        a = self._z[key]
        if a.dtype.kind in "US":
            return a
        return torch.from_numpy(a)

    def image(self, key):
        """uint8 image fixture -> [1,1,H,W] float32 in [0,1]."""
        a = self._z[key]
        t = torch.from_numpy(a.astype(np.float32) / 255)
        while t.dim() < 4:
            t = t[None]
        return t


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()

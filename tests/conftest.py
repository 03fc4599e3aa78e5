import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# The CPU oracle (torch-CPU) is the checker of most GPU tests.  On the GPU box torch starts 128 threads on a 16-CPU share
# of a 256-CPU host, and the oracle's matcher then takes 1.9 s per VGA pair instead of 0.43 s with 16 threads
# (profiles/r06_oracle_threads.txt): bound the pool here and, through the environment, in the child processes tests start.
_THREADS = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1))
torch.set_num_threads(_THREADS)
os.environ.setdefault("OMP_NUM_THREADS", str(_THREADS))


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

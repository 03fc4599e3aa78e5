"""Randomised shape sweep of the hand-written kernels with ragged edges (GEMM epilogues at any M x N, attention problem
sets, Winograd convolution at any H x W / channel
count / pooling, the DISK window-NMS + top-n selection, the assignment head at any m x n) against torch float64 and the
oracle; the sweep itself is tools/micro/fuzz_shapes.py so that it can be run for longer by hand."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "micro"))


def test_ragged_shapes_sweep():
    import fuzz_shapes
    assert fuzz_shapes.run(n_cases=40, seed=7) == []

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


def test_module_level_sweep_vs_oracle():
    """The extractor / matcher MODULES on random configurations (odd image sizes, every NMS radius, borders, thresholds,
    top-k above / below the candidate count, RGB, official sampling modes with image_size; tiny / unequal / batched
    key-point sets) against the oracle, which tests/test_oracle_vs_reference_fuzz.py pins to the reference on the same
    kind of configurations."""
    import fuzz_models
    assert fuzz_models.run(n_cases=16, seed=5) == []

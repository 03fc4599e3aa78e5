"""CPU oracle for the SuperPoint + LightGlue hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU (fp32) restatement of
the reference's algorithm for the path named by BASELINE.json's north_star.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it, and there only as the checker / the timed CPU baseline -- never as a fallback of
the product path (`glue_factory_colon_amd` fails loudly when its HIP library is
missing and never imports `oracle`).

Pinning: every function here is checked in `tests/test_oracle_golden.py` against
golden vectors produced by running the reference's own modules from /root/reference
in the build container (`tests/golden/make_golden.py`, committed together with the
vectors).  The reference publishes no tensor-level fixtures of its own.

Each function cites the reference file:line it restates (paths relative to the
reference repository root).
"""
from . import eval_homography, lightglue, preprocess, superpoint  # noqa: F401

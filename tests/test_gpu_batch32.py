"""BASELINE.json configs[1] at the batch bench.py times: 32 VGA pairs, 1024 key points, through extractor and
matcher exactly as `bench.py::step` does (both views in one extractor call of 64 images, `force_num_keypoints`).

At this size the library dispatches, on its own, to the kernel variants the benchmark measures
(gemm_nt_kernel<2,2,16>, attention_kernel<2,4> / the shared-sim cross kernel, the row-owning FFN GEMM, the
two-pass assignment tail); smaller test batches never reach them.  Checked here:
  * ALL 32 pairs against the CPU oracle (reference path restated, oracle/): key-point sets, matched
    coordinate pairs, scores;
  * batch invariance: all 32 pairs bit-identical (every output tensor) to the same pairs run 8 at a time, and
    integer outputs identical / scores within 1e-4 (measured 1.9e-5) of the same pairs run 2 at a time (the small-batch path splits
    the attention keys over workgroups, so floats may differ in the last bits there).
Reference: gluefactory/models/matchers/lightglue.py:422-553, extractors/superpoint_open.py:126-232.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from glue_factory_colon_amd import lightglue, superpoint_open, synthetic, weights  # noqa: E402
from oracle import lightglue as olg  # noqa: E402
from oracle import superpoint as osp  # noqa: E402
from parity_utils import compare_keypoints, match_pairs, record  # noqa: E402

DEV = "cuda"
H, W, K, B = 480, 640, 1024, 32
# every pair of the benchmarked batch (the CPU oracle does ~0.4 pairs/s on the GPU box's host share: ~90 s); a child run
# of this test under another build of the library (test_exact_erf_build_...) checks a spread of four
_N_ORACLE = int(os.environ.get("GFC_TEST_ORACLE_PAIRS", B))
ORACLE_PAIRS = tuple(range(B)) if _N_ORACLE >= B else tuple(sorted({round(i * (B - 1) / max(_N_ORACLE - 1, 1)) for i in range(_N_ORACLE)}))


def run_batch(ext, mat, v0, v1):
    """bench.py::step on a list of pairs."""
    b = v0.shape[0]
    size = torch.tensor([[float(W), float(H)]] * b, device=DEV)
    torch.manual_seed(7)  # pad_random_c (only used when an image has fewer than K detections)
    pj = ext({"image": torch.cat([v0, v1], 0), "image_size": torch.cat([size, size], 0)})
    p0 = {k: v[:b] for k, v in pj.items()}
    p1 = {k: v[b:] for k, v in pj.items()}
    out = mat({"keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"], "descriptors0": p0["descriptors"],
               "descriptors1": p1["descriptors"], "view0": {"image_size": size}, "view1": {"image_size": size}})
    return p0, p1, out


@pytest.fixture(scope="module")
def c2_batch32():
    v0, v1 = synthetic.synthetic_pairs(B, H, W, seed=1234, device=DEV)  # bench.py rank 0 inputs
    ext = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": K, "detection_threshold": 0.0,
                                      "nms_radius": 3, "force_num_keypoints": True}).eval().to(DEV)
    mat = lightglue.LightGlue({"weights": "synthetic", "filter_threshold": 0.1, "depth_confidence": -1,
                               "width_confidence": -1}).eval().to(DEV)
    with torch.no_grad():
        p0, p1, out = run_batch(ext, mat, v0, v1)
    torch.cuda.synchronize()
    return v0, v1, ext, mat, p0, p1, out


def test_c2_batch32_vs_oracle(c2_batch32):
    v0, v1, _, _, p0, p1, out = c2_batch32
    sd_sp, sd_lg = weights.superpoint_open_state_dict(0), weights.lightglue_state_dict(0)
    size = torch.tensor([[float(W), float(H)]])
    n_ref_total = n_same = n_elementwise = 0
    worst = 0.0
    for i in ORACLE_PAIRS:
        imgs = torch.cat([v0[i:i + 1], v1[i:i + 1]], 0).cpu()
        o = osp.extract(sd_sp, imgs, "open", nms_radius=3, max_num_keypoints=K, detection_threshold=0.0)
        okp, osc, ode = torch.stack(o["keypoints"]), torch.stack(o["keypoint_scores"]), torch.stack(o["descriptors"])
        assert okp.shape == (2, K, 2)  # every image of this workload has more than K detections: no padding
        for side, p in ((0, p0), (1, p1)):
            compare_keypoints(f"c2_b32_pair{i}_view{side}", p["keypoints"][i], p["keypoint_scores"][i],
                              p["descriptors"][i], okp[side], osc[side], ode[side], radius=3)
        ref = olg.match(sd_lg, okp[:1], okp[1:], ode[:1], ode[1:], size, size, filter_threshold=0.1)
        mine = match_pairs(p0["keypoints"][i], p1["keypoints"][i], out["matches0"][i])
        theirs = match_pairs(okp[0], okp[1], ref["matches0"][0])
        n_ref_total += len(theirs)
        n_same += len(mine & theirs)
        assert len(theirs) > 500
        assert mine == theirs, (i, len(mine), len(theirs), len(mine ^ theirs))  # measured: identical since round 2
        # scores of the common matched pairs
        def by_pair(kp0, kp1, m0, s0):
            kp0, kp1, m0, s0 = kp0.cpu(), kp1.cpu(), m0.cpu(), s0.cpu()
            return {(*kp0[a].tolist(), *kp1[int(m0[a])].tolist()): float(s0[a]) for a in (m0 >= 0).nonzero().flatten().tolist()}
        sm = by_pair(p0["keypoints"][i], p1["keypoints"][i], out["matches0"][i], out["matching_scores0"][i])
        sr = by_pair(okp[0], okp[1], ref["matches0"][0], ref["matching_scores0"][0])
        err = max(abs(sm[q] - sr[q]) for q in set(sm) & set(sr))
        worst = max(worst, err)
        assert err < 1e-4, (i, err)
        # element-wise view: when no near-tie swapped two ranks of the top-k order in either view, the key-point ARRAYS
        # are the oracle's and then matches0 / matches1 must be too, index by index
        if torch.equal(p0["keypoints"][i].cpu(), okp[0]) and torch.equal(p1["keypoints"][i].cpu(), okp[1]):
            assert torch.equal(out["matches0"][i].cpu(), ref["matches0"][0]), i
            assert torch.equal(out["matches1"][i].cpu(), ref["matches1"][0]), i
            n_elementwise += 1
    record("c2_batch32_vs_oracle", pairs=len(ORACLE_PAIRS), ref_matches=n_ref_total, identical=n_same, score_err=worst,
           pairs_with_elementwise_identical_arrays=n_elementwise)


def test_c2_batch32_batch_invariance(c2_batch32):
    v0, v1, ext, mat, p0, p1, out = c2_batch32
    keys_ext = ("keypoints", "keypoint_scores", "descriptors")
    keys_mat = ("matches0", "matches1", "matching_scores0", "matching_scores1", "log_assignment", "ref_descriptors0",
                "ref_descriptors1")
    # every image had >= K detections (no random padding), otherwise the comparison below would depend on the RNG
    assert (p0["keypoint_scores"] > 0).all() and (p1["keypoint_scores"] > 0).all()
    with torch.no_grad():
        for s in range(0, B, 8):  # 8 at a time: other GEMM tile (64x64), other attention variant (<1,4>), no key split
            q0, q1, o8 = run_batch(ext, mat, v0[s:s + 8], v1[s:s + 8])
            for k in keys_ext:
                assert torch.equal(q0[k], p0[k][s:s + 8]) and torch.equal(q1[k], p1[k][s:s + 8]), (s, k)
            for k in keys_mat:
                assert torch.equal(o8[k], out[k][s:s + 8]), (s, k)
        worst = 0.0
        for s in (0, 14, 30):  # 2 at a time: the small-batch path (attention keys split over workgroups + merge)
            q0, q1, o2 = run_batch(ext, mat, v0[s:s + 2], v1[s:s + 2])
            for k in keys_ext:
                assert torch.equal(q0[k], p0[k][s:s + 2]) and torch.equal(q1[k], p1[k][s:s + 2]), (s, k)
            assert torch.equal(o2["matches0"], out["matches0"][s:s + 2]), s
            assert torch.equal(o2["matches1"], out["matches1"][s:s + 2]), s
            worst = max(worst, float((o2["matching_scores0"] - out["matching_scores0"][s:s + 2]).abs().max()))
        assert worst < 1e-4, worst  # measured 1.9e-5: key-split soft-max partials are merged in another order
    record("c2_batch32_batch_invariance", score_diff_vs_batch2=worst)

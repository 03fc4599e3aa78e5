"""BASELINE.json configs[1] at the batch bench.py times: 32 VGA pairs, 1024 key points, through extractor and
matcher exactly as `bench.py::step` does (both views in one extractor call of 64 images, `force_num_keypoints`).

At this size the library dispatches, on its own, to the kernel variants the benchmark measures
(gemm_nt_kernel<2,2,16>, attention_kernel<2,4> / the shared-sim cross kernel, the row-owning FFN GEMM, the
two-pass assignment tail); smaller test batches never reach them.  Checked here:
  * ALL 32 pairs against THE REFERENCE ITSELF (tests/golden/c2_batch32.npz: the reference's TwoViewPipeline run on
    the same 32 pairs in the build container, make_golden.py::golden_c2_batch32): key points, scores, descriptors,
    and matches0 / matches1 / matching scores index by index through the key-point correspondence;
  * the matcher stage-isolated AT THIS BATCH: the batch-32 matcher call's own inputs (the HIP extractor's features of
    all 32 pairs) through the CPU oracle's matcher -> `torch.equal(matches0 / matches1)`, scores and log-assignment
    within 1e-4 -- the element-wise index check on exactly the kernels the benchmark dispatches;
  * 4 pairs end to end against the live CPU oracle (reference path restated, oracle/; all 32 with
    GFC_TEST_ORACLE_PAIRS=32);
  * batch invariance: all 32 pairs bit-identical (every output tensor) to the same pairs run 8 at a time, and
    integer outputs identical / scores within 1e-4 (measured 1.9e-5) of the same pairs run 2 at a time (the small-batch path splits
    the attention keys over workgroups, so floats may differ in the last bits there).
Reference: gluefactory/models/matchers/lightglue.py:294-319,422-553, extractors/superpoint_open.py:126-232,
two_view_pipeline.py:278-339.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from glue_factory_colon_amd import lightglue, superpoint_open, synthetic, weights  # noqa: E402
from oracle import lightglue as olg  # noqa: E402
from oracle import superpoint as osp  # noqa: E402
from parity_utils import compare_keypoints, compare_with_reference_pair, image_sha256, match_pairs, record  # noqa: E402

DEV = "cuda"
H, W, K, B = 480, 640, 1024, 32
# live end-to-end oracle: a spread of four pairs by default (the CPU oracle does ~0.4 pairs/s on the GPU box's host share;
# all 32 pairs are compared with the reference-made fixture instead), GFC_TEST_ORACLE_PAIRS=32 for every pair
_N_ORACLE = int(os.environ.get("GFC_TEST_ORACLE_PAIRS", 4))
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
    n_ref_total = n_same = 0
    worst = 0.0
    for i in ORACLE_PAIRS:
        imgs = torch.cat([v0[i:i + 1], v1[i:i + 1]], 0).cpu()
        o = osp.extract(sd_sp, imgs, "open", nms_radius=3, max_num_keypoints=K, detection_threshold=0.0)
        okp, osc, ode = torch.stack(o["keypoints"]), torch.stack(o["keypoint_scores"]), torch.stack(o["descriptors"])
        assert okp.shape == (2, K, 2)  # every image of this workload has more than K detections: no padding
        for side, p in ((0, p0), (1, p1)):
            compare_keypoints(f"c2_b32_oracle_pair{i}_view{side}", p["keypoints"][i], p["keypoint_scores"][i],
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
    record("c2_batch32_vs_oracle", pairs=len(ORACLE_PAIRS), ref_matches=n_ref_total, identical=n_same, score_err=worst)


def test_c2_batch32_vs_reference_fixture(c2_batch32, golden):
    """All 32 pairs of the benchmarked batch against what the reference itself produced for them."""
    v0, v1, _, _, p0, p1, out = c2_batch32
    g = golden("c2_batch32")
    assert g["pair_ids"].tolist() == list(range(B))
    n_ref = n_idx = 0
    worst = 0.0
    for i in range(B):
        assert image_sha256(v0[i], v1[i]) == str(g["image_sha256"][i]), i  # the fixture's inputs ARE these images
        r, n, w = compare_with_reference_pair(f"c2_b32_pair{i}", g, i, p0, p1, out, i, radius=3, min_matches=500)
        n_ref, n_idx, worst = n_ref + r, n_idx + n, max(worst, w)
    record("c2_batch32_vs_reference", pairs=B, ref_matches=n_ref, indices_compared=n_idx, indices_identical=n_idx,
           matching_score_err=worst)


def test_c2_batch32_matcher_stage_isolated(c2_batch32):
    """The matcher call the benchmark makes (ONE call, 32 pairs: attention_kernel<2,4>, gemm_nt_kernel<2,2,16,2>,
    gemm_rows512_ln_gelu_kernel<2,true>, two-pass assignment tail), element-wise: its own inputs through the CPU
    oracle's matcher must give the same matches0 / matches1, index by index (lightglue.py:294-319)."""
    _, _, _, _, p0, p1, out = c2_batch32
    sd_lg = weights.lightglue_state_dict(0)
    size = torch.tensor([[float(W), float(H)]])
    identical, n_matches = 0, 0
    s_err = la_err = 0.0
    chunk = 8  # pairs per oracle call (bounds the oracle's [chunk, 4, K, K] attention tensors)
    for s in range(0, B, chunk):
        e = s + chunk
        ref = olg.match(sd_lg, p0["keypoints"][s:e].cpu(), p1["keypoints"][s:e].cpu(), p0["descriptors"][s:e].cpu(),
                        p1["descriptors"][s:e].cpu(), size.expand(chunk, 2), size.expand(chunk, 2), filter_threshold=0.1)
        for key in ("matches0", "matches1"):
            assert out[key].dtype == torch.long
            assert torch.equal(out[key][s:e].cpu(), ref[key]), (s, key, int((out[key][s:e].cpu() != ref[key]).sum()))
        for key in ("matching_scores0", "matching_scores1"):
            s_err = max(s_err, float((out[key][s:e].cpu() - ref[key]).abs().max()))
        la = ref["log_assignment"]
        la_err = max(la_err, float(((out["log_assignment"][s:e].cpu() - la).abs() / (1 + la.abs())).max()))
        identical += chunk
        n_matches += int((ref["matches0"] >= 0).sum())
    assert n_matches > 500 * B
    assert s_err < 1e-4 and la_err < 1e-4, (s_err, la_err)
    record("c2_batch32_matcher_stage_isolated", matcher_stage_isolated_identical=f"{identical}/{B}", matches=n_matches,
           matching_score_err=s_err, log_assignment_rel_err=la_err)


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

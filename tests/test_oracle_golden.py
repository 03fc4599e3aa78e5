"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference's own
modules (tests/golden/make_golden.py).  Indices / counts bit-exact, floats <= 1e-5
(the north-star tolerance for the HIP path is 1e-4; the oracle itself sits much closer)."""
import pytest
import torch

from glue_factory_colon_amd import weights
from oracle import lightglue as olg
from oracle import superpoint as osp

TOL = 1e-5


def close(a, b, tol=TOL):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    assert err <= tol, err


@pytest.mark.parametrize("r", [0, 1, 3, 4])
def test_nms_bit_exact(golden, r):
    g = golden("nms")
    out = osp.nms(g[f"in_r{r}"], r)
    assert torch.equal(out, g[f"out_r{r}"])
    assert torch.equal(out, g[f"out_official_r{r}"])


def test_assignment_and_filter(golden):
    g = golden("assignment")
    la = olg.log_double_softmax(g["sim"], g["z0"], g["z1"])
    close(la, g["log_assignment"], 1e-5)
    for th, tag in ((0.0, "0p0"), (0.1, "0p1"), (0.5, "0p5")):
        m0, m1, s0, s1 = olg.filter_matches(g["log_assignment"], th)
        assert torch.equal(m0, g[f"m0_{tag}"]) and torch.equal(m1, g[f"m1_{tag}"])
        assert torch.equal(s0, g[f"s0_{tag}"]) and torch.equal(s1, g[f"s1_{tag}"])
    m0, m1, s0, s1 = olg.filter_matches(g["perm_log_assignment"], 0.1)
    assert torch.equal(m0, g["perm_m0"]) and torch.equal(m1, g["perm_m1"])
    assert (m0 >= 0).all()
    m0, m1, s0, s1 = olg.filter_matches(torch.zeros((2, 1, 6)), 0.1)
    assert torch.equal(m0, g["empty_m0"]) and torch.equal(m1, g["empty_m1"])
    assert m0.shape == (2, 0) and m1.shape == (2, 5) and m1.dtype == torch.int64


def test_superpoint_open_stages(golden):
    g = golden("superpoint_open")
    sd = weights.superpoint_open_state_dict(0)
    heat, dense = osp.dense_open(sd, g["image"])
    close(heat, g["heatmap"], 1e-6)
    close(dense[0], g["dense_desc_0"], 1e-5)
    close(osp.logits_to_heatmap(g["logits"]), g["heatmap"], 1e-7)
    assert torch.equal(osp.nms(g["heatmap"], 3), g["nms_r3"])


def test_superpoint_open_outputs(golden):
    g = golden("superpoint_open")
    sd = weights.superpoint_open_state_dict(0)
    out = osp.extract(sd, g["image"], "open", nms_radius=3, max_num_keypoints=150, detection_threshold=0.0)
    for i in range(2):
        assert torch.equal(out["keypoints"][i], g[f"k150_kpts_{i}"])
        close(out["keypoint_scores"][i], g[f"k150_scores_{i}"], 1e-6)
        close(out["descriptors"][i], g[f"k150_desc_{i}"])
    out = osp.extract(sd, g["image"][:1], "open", nms_radius=4, max_num_keypoints=4096, detection_threshold=0.0)
    assert out["keypoints"][0].shape[0] < 4096  # fewer than k: all candidates, row-major order
    assert torch.equal(out["keypoints"][0], g["k4096_r4_kpts_0"])
    close(out["descriptors"][0], g["k4096_r4_desc_0"])
    out = osp.extract(sd, g["image_rgb"], "open", nms_radius=0, max_num_keypoints=100, detection_threshold=0.02,
                      remove_borders=6)
    assert torch.equal(out["keypoints"][0], g["rgb_r0_kpts"])
    close(out["keypoint_scores"][0], g["rgb_r0_scores"], 1e-6)
    close(out["descriptors"][0], g["rgb_r0_desc"])
    out = osp.extract(sd, g["image"], "open", nms_radius=3, max_num_keypoints=64, detection_threshold=0.0)
    assert torch.equal(torch.stack(out["keypoints"]), g["b2_k64_kpts"])
    close(torch.stack(out["descriptors"]), g["b2_k64_desc"])


def test_specular_mask_filter(golden):
    """The reference's Endomapper addition: open variant filters before top-k, official variant after."""
    g = golden("specular")
    sd = weights.superpoint_open_state_dict(0)
    mask = g["mask"].bool()
    out = osp.extract(sd, g["image"], "open", nms_radius=3, max_num_keypoints=150, detection_threshold=0.0,
                      specular_mask=mask)
    for i in range(2):
        assert out["keypoints"][i].shape[0] == 150  # filtered first, then top-k: still k key points
        assert torch.equal(out["keypoints"][i], g[f"open_k150_kpts_{i}"])
        close(out["keypoint_scores"][i], g[f"open_k150_scores_{i}"], 1e-6)
        close(out["descriptors"][i], g[f"open_k150_desc_{i}"])
        xy = (out["keypoints"][i] - 0.5).long()
        assert mask[i, 0][xy[:, 1], xy[:, 0]].all()
    out = osp.extract(sd, g["image"][:1], "open", nms_radius=3, max_num_keypoints=150, detection_threshold=0.0,
                      specular_mask=mask[:1, 0].float(), image_size=g["open_crop_size"])
    assert torch.equal(out["keypoints"][0], g["open_crop_kpts"])
    assert out["keypoints"][0][:, 0].max() < 131 and out["keypoints"][0][:, 1].max() < 97
    out = osp.extract(sd, g["image"], "open", nms_radius=3, max_num_keypoints=48, detection_threshold=0.0,
                      specular_mask=mask)
    assert torch.equal(torch.stack(out["keypoints"]), g["open_b2_k48_kpts"])
    sdo = weights.superpoint_state_dict(0)
    out = osp.extract(sdo, g["image"], "official", nms_radius=3, max_num_keypoints=150, detection_threshold=0.0005,
                      specular_mask=mask)
    for i in range(2):
        assert out["keypoints"][i].shape[0] < 150  # top-k first, then filtered
        assert torch.equal(out["keypoints"][i], g[f"off_k150_kpts_{i}"])
        close(out["keypoint_scores"][i], g[f"off_k150_scores_{i}"], 1e-6)
        close(out["descriptors"][i], g[f"off_k150_desc_{i}"])


def test_soft_argmax_refinement(golden):
    g = golden("specular")
    sdo = weights.superpoint_state_dict(0)
    out = osp.extract(sdo, g["image"][:1], "official", nms_radius=3, max_num_keypoints=150, detection_threshold=0.0005,
                      refinement_radius=2)
    close(out["keypoints"][0], g["refine_kpts"], 1e-5)
    assert (out["keypoints"][0] - out["keypoints"][0].round()).abs().max() > 0.05  # really fractional
    close(out["descriptors"][0], g["refine_desc"])
    out = osp.extract(sdo, g["image"][1:2], "official", nms_radius=3, max_num_keypoints=150, detection_threshold=0.0005,
                      refinement_radius=2, specular_mask=g["mask"][1:2].bool())
    assert out["keypoints"][0].shape == g["refine_spec_kpts"].shape
    close(out["keypoints"][0], g["refine_spec_kpts"], 1e-5)


def test_superpoint_official(golden):
    g = golden("superpoint_official")
    sd = weights.superpoint_state_dict(0)
    heat, dense = osp.dense_official(sd, g["image"])
    close(heat, g["heatmap"], 1e-6)
    close(dense, g["dense_desc"], 1e-5)
    for legacy, tag in ((True, "legacy"), (False, "fixed")):
        out = osp.extract(sd, g["image"], "official", nms_radius=3, max_num_keypoints=120,
                          detection_threshold=0.0, legacy_sampling=legacy)
        assert torch.equal(out["keypoints"][0], g[f"{tag}_kpts"])
        close(out["keypoint_scores"][0], g[f"{tag}_scores"], 1e-6)
        close(out["descriptors"][0], g[f"{tag}_desc"])
    out = osp.extract(sd, g["image"], "official", nms_radius=4, max_num_keypoints=-1, detection_threshold=0.01,
                      image_size=g["sized_image_size"])
    assert torch.equal(out["keypoints"][0], g["sized_kpts"])
    close(out["descriptors"][0], g["sized_desc"])
    assert out["keypoints"][0][:, 0].max() < 120 - 4 + 0.5 and out["keypoints"][0][:, 1].max() < 90 - 4 + 0.5


def test_lightglue_layers_and_matches(golden):
    g = golden("lightglue")
    sd = weights.lightglue_state_dict(0)
    k0 = olg.normalize_keypoints(g["keypoints0"], g["image_size"])
    e0 = olg.positional_encoding(sd["posenc.Wr.weight"], k0)
    close(e0, g["enc0"], 1e-6)
    close(olg.self_block(sd, "transformers.0.self_attn", g["descriptors0"], e0, 4), g["layer0_self0"])
    out = olg.match(sd, g["keypoints0"], g["keypoints1"], g["descriptors0"], g["descriptors1"], g["image_size"],
                    g["image_size"], filter_threshold=0.1, return_layers=True)
    close(out["layers"][0][0], g["layer0_desc0"], 2e-5)
    close(out["layers"][4][1], g["layer4_desc1"], 5e-5)
    close(out["ref_descriptors0"], g["b2_ref_descriptors0"], 1e-4)
    close(out["log_assignment"], g["b2_log_assignment"], 1e-3)
    assert torch.equal(out["matches0"], g["b2_matches0"]) and torch.equal(out["matches1"], g["b2_matches1"])
    assert (out["matches0"] >= 0).sum() > 30  # non-trivial assignment
    close(out["matching_scores0"], g["b2_matching_scores0"], 1e-4)
    close(out["prune0"], g["b2_prune0"], 0)
    out = olg.match(sd, g["keypoints0"], g["keypoints1"], g["descriptors0"], g["descriptors1"], g["image_size"],
                    g["image_size"], filter_threshold=0.0)
    assert torch.equal(out["matches0"], g["th0_matches0"]) and torch.equal(out["matches1"], g["th0_matches1"])


def test_lightglue_ragged_and_128d(golden):
    g = golden("lightglue")
    sd = weights.lightglue_state_dict(0)
    size = g["image_size"][:1]
    out = olg.match(sd, g["keypoints0"][:1, :100], g["keypoints1"][:1], g["descriptors0"][:1, :100],
                    g["descriptors1"][:1], size, size, filter_threshold=0.1)
    assert out["log_assignment"].shape == (1, 101, 161)
    assert torch.equal(out["matches0"], g["ragged_matches0"]) and torch.equal(out["matches1"], g["ragged_matches1"])
    close(out["matching_scores1"], g["ragged_matching_scores1"], 1e-4)
    sd128 = weights.lightglue_state_dict(0, input_dim=128)
    out = olg.match(sd128, g["d128_keypoints0"], g["d128_keypoints1"], g["d128_descriptors0"],
                    g["d128_descriptors1"], size, size, filter_threshold=0.1)
    assert torch.equal(out["matches0"], g["d128_matches0"]) and torch.equal(out["matches1"], g["d128_matches1"])
    close(out["log_assignment"], g["d128_log_assignment"], 1e-3)


def test_lightglue_add_scale_ori(golden):
    g = golden("scale_ori")
    sd = weights.lightglue_state_dict(0, add_scale_ori=True)
    assert sd["posenc.Wr.weight"].shape == (32, 4)
    so0 = torch.cat([g["scales0"][..., None], g["oris0"]], -1)            # oris0 came as [B,K,1]
    so1 = torch.stack([g["scales1"], g["oris1"]], -1)
    out = olg.match(sd, g["keypoints0"], g["keypoints1"], g["descriptors0"], g["descriptors1"], g["image_size"],
                    g["image_size"], filter_threshold=0.1, scale_ori0=so0, scale_ori1=so1)
    assert torch.equal(out["matches0"], g["matches0"]) and torch.equal(out["matches1"], g["matches1"])
    close(out["matching_scores0"], g["matching_scores0"], 1e-5)
    close(out["ref_descriptors0"], g["ref_descriptors0"], 1e-4)
    assert int((g["matches0"] >= 0).sum()) > 40  # a real assignment, not the empty one


def test_lightglue_adaptive_pruning(golden):
    """Point pruning (and the depth check on a run that reaches the last layer) against the reference."""
    g = golden("lightglue_adaptive")
    sd = weights.lightglue_adaptive_state_dict(0, prune_z=1.5)  # as make_golden.py (ADAPTIVE_PRUNE_Z)
    size = g["image_size"]
    for tag, conf in (("prune", dict(width_confidence=0.95)), ("both", dict(width_confidence=0.95, depth_confidence=0.95))):
        out = olg.match_adaptive(sd, g["keypoints0"], g["keypoints1"], g["descriptors0"], g["descriptors1"], size, size,
                                 filter_threshold=0.1, **conf)
        assert out["log_assignment"].shape == g[f"{tag}_log_assignment"].shape  # same surviving point counts
        assert out["log_assignment"].shape[1] < 513 and out["stop_layer"] == 9
        for key in ("matches0", "matches1", "prune0", "prune1"):
            assert torch.equal(out[key], g[f"{tag}_{key}"]), (tag, key)
        close(out["matching_scores0"], g[f"{tag}_matching_scores0"], 1e-4)
        close(out["log_assignment"], g[f"{tag}_log_assignment"], 1e-3)


def test_nn_matcher(golden):
    g = golden("nn_matcher")
    for tag, conf in (("default", {}), ("ratio", {"ratio_thresh": 0.9}), ("dist", {"distance_thresh": 0.9}),
                      ("nomutual", {"mutual_check": False, "ratio_thresh": 0.95, "distance_thresh": 1.1})):
        out = olg.nn_match(g["descriptors0"], g["descriptors1"], **conf)
        for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
            assert torch.equal(out[key], g[f"{tag}_{key}"]), (tag, key)
    close(out["similarity"], g["similarity"], 1e-6)
    close(olg.nn_match(g["descriptors0"], g["descriptors1"])["log_assignment"], g["log_assignment"], 1e-5)


def test_disk_oracle_known_answers():
    """oracle/disk.py restates kornia's detection functions (kornia is absent: parity unpinned).  Hand-checkable cases:
    first-maximum tie rule of max_pool2d(return_indices), strict cutoff, the (n+1)-th-score threshold (which drops the
    minimum when there are not more than n candidates), row-major output order, descriptor normalisation."""
    from oracle import disk as odisk

    s = torch.full((1, 1, 7, 9), -1.0)
    s[0, 0, 2, 2] = 0.5
    s[0, 0, 2, 3] = 0.5      # tie inside one window: only the first (row-major) maximum survives
    s[0, 0, 5, 7] = 0.25
    s[0, 0, 0, 8] = 0.0      # not > cutoff 0
    keep = odisk.window_nms(s.squeeze(1), 5, 0.0)
    assert keep.nonzero().tolist() == [[0, 2, 2], [0, 5, 7]]
    (xy, sc), = odisk.heatmap_to_keypoints(s, None, 5, 0.0)
    assert xy.tolist() == [[2, 2], [7, 5]] and sc.tolist() == [0.5, 0.25]
    (xy, sc), = odisk.heatmap_to_keypoints(s, 1, 5, 0.0)      # 2 candidates > n = 1: the best one
    assert xy.tolist() == [[2, 2]]
    (xy, sc), = odisk.heatmap_to_keypoints(s, 2, 5, 0.0)      # 2 candidates <= n: the minimum is dropped (kornia)
    assert xy.tolist() == [[2, 2]]
    dense = torch.zeros((4, 7, 9))
    dense[:, 2, 2] = torch.tensor([3.0, 0.0, 4.0, 0.0])
    d = odisk.merge_with_descriptors(torch.tensor([[2, 2], [0, 0]]), dense)
    assert torch.allclose(d[0], torch.tensor([0.6, 0.0, 0.8, 0.0])) and (d[1] == 0).all()


def test_boat_pair_native_size(golden):
    """BASELINE config 1 on the oracle: assets/boat1.png <-> boat2.png at 850 x 680 (the reference's TwoViewPipeline on
    its CPU path produced the vectors): key points bit-exact, scores <= 1e-5, the (empty) match set at filter_threshold
    0.1 and the 10 mutual matches at filter_threshold 0."""
    g = golden("boat_native")
    sd = weights.superpoint_open_state_dict(0)
    feats = []
    for i in "01":
        img = (g["image" + i].float() / 255).permute(2, 0, 1)[None].contiguous()
        o = osp.extract(sd, img, "open", nms_radius=3, max_num_keypoints=1024, detection_threshold=0.0)
        assert torch.equal(o["keypoints"][0], g["keypoints" + i][0])
        close(o["keypoint_scores"][0], g["keypoint_scores" + i][0], 1e-6)
        feats.append(o)
    size = torch.tensor([[850.0, 680.0]])
    args = (weights.lightglue_state_dict(0), torch.stack(feats[0]["keypoints"]), torch.stack(feats[1]["keypoints"]),
            torch.stack(feats[0]["descriptors"]), torch.stack(feats[1]["descriptors"]), size, size)
    for th, tag in ((0.1, ""), (0.0, "th0_")):
        out = olg.match(*args, filter_threshold=th)
        assert torch.equal(out["matches0"], g[tag + "matches0"]) and torch.equal(out["matches1"], g[tag + "matches1"])
        close(out["matching_scores0"], g[tag + "matching_scores0"])
        close(out["matching_scores1"], g[tag + "matching_scores1"])
    assert int((g["matches0"] >= 0).sum()) == 0 and int((g["th0_matches0"] >= 0).sum()) == 10


def test_official_pipeline_ragged_counts_c3(golden):
    """BASELINE config 3 on the oracle, the official extractor's RAGGED path: with detection_threshold 0.5 the views keep
    fewer than 1024 key points, different numbers per view (reference vectors of the reference's own TwoViewPipeline,
    make_golden.py::golden_pipeline_official_ragged; records as utils/export_predictions.py:36-85 writes them, key
    points divided by `scales`).  Two of the five pairs (CPU time): counts and indices bit-exact, key-point scores <= 1e-6, matching scores <= 3e-5."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from c3_inputs import C3_RAGGED_PAIRS, C3_RAGGED_THRESHOLD, c3_pair

    g = golden("pipeline_official_ragged")
    sd_sp, sd_lg = weights.superpoint_state_dict(0), weights.lightglue_state_dict(0)
    for i in (1, 3):  # (956, 941) and (944, 1024): below the cap in both views / capped in one
        name, seed, s0, s1, origs = C3_RAGGED_PAIRS[i]
        data = c3_pair(seed, s0, s1, origs)
        feats = []
        for v in "01":
            view = data["view" + v]
            o = osp.extract(sd_sp, view["image"], "official", nms_radius=3, max_num_keypoints=1024,
                            detection_threshold=C3_RAGGED_THRESHOLD, image_size=view["image_size"])
            kp = o["keypoints"][0] * (1.0 / view["scales"])
            assert kp.shape == g[f"p{i}_keypoints{v}"].shape, (name, v, kp.shape)
            close(kp, g[f"p{i}_keypoints{v}"], 1e-4)  # original-image pixels (values up to ~1300)
            close(o["keypoint_scores"][0], g[f"p{i}_keypoint_scores{v}"], 1e-6)
            feats.append(o)
        out = olg.match(sd_lg, torch.stack(feats[0]["keypoints"]), torch.stack(feats[1]["keypoints"]),
                        torch.stack(feats[0]["descriptors"]), torch.stack(feats[1]["descriptors"]),
                        data["view0"]["image_size"], data["view1"]["image_size"], filter_threshold=0.1)
        assert torch.equal(out["matches0"][0], g[f"p{i}_matches0"]) and torch.equal(out["matches1"][0], g[f"p{i}_matches1"])
        # ~950 x ~950 points through 9 layers: the oracle's own restatement of the attention / assignment sums differs
        # from the reference's fused SDPA by up to 1.3e-5 here (measured); the bar of the HIP path is 1e-4
        close(out["matching_scores0"][0], g[f"p{i}_matching_scores0"], 3e-5)
        close(out["matching_scores1"][0], g[f"p{i}_matching_scores1"], 3e-5)
        assert int((g[f"p{i}_matches0"] >= 0).sum()) > 300


@pytest.mark.parametrize("fixture, h, w, k, entries", [("c2_batch32", 480, 640, 1024, (0, 17)), ("c4_pairs", 1024, 1024, 2048, (1,))])
def test_oracle_at_the_benchmarked_configurations(golden, fixture, h, w, k, entries):
    """The oracle against what the reference itself (TwoViewPipeline, batch 1) produced on pairs of the benchmark's own
    synthetic batches (BASELINE configs[1] and [3]; make_golden.py::golden_c2_batch32 / golden_c4_pairs): key points and
    matches bit-exact, floats <= 1e-5 (matching scores <= 5e-5).  The GPU tests compare the HIP path with the same fixtures."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from bench_inputs import desc_checksum_weights
    from glue_factory_colon_amd import synthetic
    from parity_utils import image_sha256

    g = golden(fixture)
    v0, v1 = synthetic.synthetic_pairs(32, h, w, seed=1234)
    sd_sp, sd_lg = weights.superpoint_open_state_dict(0), weights.lightglue_state_dict(0)
    size = torch.tensor([[float(w), float(h)]])
    for j in entries:
        i = int(g["pair_ids"][j])
        assert image_sha256(v0[i], v1[i]) == str(g["image_sha256"][j])
        o = osp.extract(sd_sp, torch.cat([v0[i:i + 1], v1[i:i + 1]]), "open", nms_radius=3, max_num_keypoints=k,
                        detection_threshold=0.0)
        for s in (0, 1):
            assert torch.equal(o["keypoints"][s], g["keypoints"][j, s].float() + 0.5)
            close(o["keypoint_scores"][s], g["keypoint_scores"][j, s], 1e-6)
            close(o["descriptors"][s][g["desc_rows"].long()], g["desc_sample"][j, s])
            close(desc_checksum_weights() @ o["descriptors"][s].T, g["desc_checksum"][j, s])
        ref = olg.match(sd_lg, o["keypoints"][0][None], o["keypoints"][1][None], o["descriptors"][0][None],
                        o["descriptors"][1][None], size, size, filter_threshold=0.1)
        for s in (0, 1):
            assert torch.equal(ref[f"matches{s}"][0], g["matches"][j, s].long())
            # nine layers deep at K >= 1024 two fp32 evaluations of the same network differ by this much in exp(score):
            # measured 1.3e-5 (C2) / 1.5e-5 (C4) between the oracle and the reference, both on torch-CPU
            close(ref[f"matching_scores{s}"][0], g["matching_scores"][j, s], 5e-5)
        assert int((ref["matches0"] >= 0).sum()) > 500

"""BASELINE config 3 as a tested path: the `superpoint+lightglue-official` HPatches run (reference
gluefactory/configs/superpoint+lightglue-official.yaml:3-13,27-33; eval/hpatches.py:98-110;
utils/export_predictions.py:36-85) -- official SuperPoint arithmetic (no BN, legacy descriptor sampling) +
LightGlue through the pretrained wrapper's configuration, RGB inputs with short side 480 and an arbitrary long side,
1024 key points, batch 1, one exported record per pair with key points in the ORIGINAL image's pixels.

Golden vectors: tests/golden/pipeline_official.npz, written in the build container by the reference's own
TwoViewPipeline + the record layout of its export loop (make_golden.py::golden_pipeline_official); inputs are
regenerated from seeds (tests/golden/c3_inputs.py).  HPatches itself is not available offline."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from c3_inputs import C3_PAIRS, C3_RAGGED_PAIRS, C3_RAGGED_THRESHOLD, c3_pair  # noqa: E402

from glue_factory_colon_amd.export_predictions import export_predictions, load_predictions  # noqa: E402
from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline  # noqa: E402
from parity_utils import record  # noqa: E402

EXPORT_KEYS = ["keypoints0", "keypoints1", "matches0", "matches1", "matching_scores0", "matching_scores1"]
OPTIONAL_KEYS = ["keypoint_scores0", "keypoint_scores1", "extractor_time_ms", "matcher_time_ms", "total_time_ms"]


def pairs_of(kp0, kp1, m0, s0):
    return {(*np.round(kp0[a], 3).tolist(), *np.round(kp1[int(m0[a])], 3).tolist()): float(s0[a])
            for a in np.nonzero(m0 >= 0)[0].tolist()}


def official_pipeline(detection_threshold=0.0, **extra):
    pipe = TwoViewPipeline({
        "extractor": {"name": "gluefactory_nonfree.superpoint", "weights": "synthetic", "max_num_keypoints": 1024,
                      "detection_threshold": detection_threshold, "nms_radius": 3},
        "matcher": {"name": "matchers.lightglue_pretrained", "features": "superpoint", "weights": "synthetic",
                    "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}, **extra}).eval()
    assert pipe.is_initialized()
    return pipe


@pytest.mark.parametrize("workers,container,pair_batch", [(1, "npz", 1), (2, "npz", 1), (1, "h5", 1), (1, "npz", 3)])
def test_c3_official_pipeline_export_golden(golden, tmp_path, workers, container, pair_batch):
    from glue_factory_colon_amd import _hdf5

    if container == "h5" and not _hdf5.available():
        pytest.skip("no HDF5 C library on this box")
    g = golden("pipeline_official")
    pipe = official_pipeline()
    loader = [{"name": [name], **c3_pair(seed, s0, s1, origs)} for name, seed, s0, s1, origs in C3_PAIRS]
    out = export_predictions(loader, pipe, tmp_path / f"predictions.{container}", keys=EXPORT_KEYS,
                             optional_keys=OPTIONAL_KEYS, workers=workers, pair_batch=pair_batch)
    recs = load_predictions(out)
    if container == "h5":  # the reference's own container (predictions.h5): HDF5 iterates names alphabetically
        assert open(out, "rb").read(4) == b"\x89HDF"
        assert sorted(recs) == sorted(n for n, *_ in C3_PAIRS)
    else:
        assert list(recs) == [n for n, *_ in C3_PAIRS] == g["names"].tolist()
    total = same = 0
    for i, (name, *_rest) in enumerate(C3_PAIRS):
        r = recs[name]
        assert set(EXPORT_KEYS) <= set(r) and r["matches0"].dtype == np.int64
        for v in "01":
            kp, ref = r["keypoints" + v], g[f"p{i}_keypoints{v}"].numpy()
            assert kp.shape == ref.shape == (1024, 2)
            # original-image pixels: (x + 0.5) / scale, not on the half-pixel grid any more
            mine = {tuple(np.round(q, 3)) for q in kp.tolist()}
            theirs = {tuple(np.round(q, 3)) for q in ref.tolist()}
            assert len(mine ^ theirs) == 0, (name, v, len(mine ^ theirs))  # measured 0 on every run since round 2
            sm = dict(zip(map(tuple, np.round(kp, 3).tolist()), r["keypoint_scores" + v].tolist()))
            sr = dict(zip(map(tuple, np.round(ref, 3).tolist()), g[f"p{i}_keypoint_scores{v}"].tolist()))
            assert max(abs(sm[q] - sr[q]) for q in set(sm) & set(sr)) < 5e-5
        pm = pairs_of(r["keypoints0"], r["keypoints1"], r["matches0"], r["matching_scores0"])
        pr = pairs_of(g[f"p{i}_keypoints0"].numpy(), g[f"p{i}_keypoints1"].numpy(), g[f"p{i}_matches0"].numpy(),
                      g[f"p{i}_matching_scores0"].numpy())
        assert len(pr) > 200
        assert set(pm) == set(pr), (name, len(pm), len(pr), len(set(pm) ^ set(pr)))  # measured: identical
        assert max(abs(pm[q] - pr[q]) for q in set(pm) & set(pr)) < 1e-4
        # matches1 is the transpose of matches0
        m0, m1 = r["matches0"], r["matches1"]
        ok = m0 >= 0
        assert (m1[m0[ok]] == np.nonzero(ok)[0]).all()
        total += len(pr)
        same += len(set(pm) & set(pr))
    record(f"c3_official_pipeline_workers{workers}_{container}_pb{pair_batch}", ref_matches=total, identical=same)


@pytest.mark.parametrize("pair_batch", [1, 5])
def test_c3_official_ragged_counts_vs_reference(golden, tmp_path, pair_batch):
    """The official extractor's ragged path at config-3 sizes (gluefactory_nonfree/superpoint.py:267-300,349-370): with
    `detection_threshold` 0.5 the views keep 852..1024 key points, different numbers in the two views of a pair.
    Reference vectors: make_golden.py::golden_pipeline_official_ragged (the reference's own TwoViewPipeline).  Counts
    are exact, key-point sets identical, matched coordinate pairs identical, scores within 1e-4 -- pair by pair
    (pair_batch 1) and with all five pairs through one ragged matcher pass (pair_batch 5)."""
    g = golden("pipeline_official_ragged")
    pipe = official_pipeline(C3_RAGGED_THRESHOLD)
    loader = [{"name": [name], **c3_pair(seed, s0, s1, origs)} for name, seed, s0, s1, origs in C3_RAGGED_PAIRS]
    out = export_predictions(loader, pipe, tmp_path / "predictions.npz", keys=EXPORT_KEYS, optional_keys=OPTIONAL_KEYS,
                             pair_batch=pair_batch)
    recs = load_predictions(out)
    assert list(recs) == [n for n, *_ in C3_RAGGED_PAIRS] == g["names"].tolist()
    total = same = 0
    counts = []
    for i, (name, *_rest) in enumerate(C3_RAGGED_PAIRS):
        r = recs[name]
        for v in "01":
            kp, ref = r["keypoints" + v], g[f"p{i}_keypoints{v}"].numpy()
            assert kp.shape == ref.shape, (name, v, kp.shape, ref.shape)  # the COUNT is exact
            mine = {tuple(np.round(q, 3)) for q in kp.tolist()}
            theirs = {tuple(np.round(q, 3)) for q in ref.tolist()}
            assert mine == theirs, (name, v, len(mine ^ theirs))
            if kp.shape[0] < 1024:  # below the cap: torch.where order (row-major), no top-k -> element-wise equal
                assert np.abs(kp - ref).max() < 1e-3
                assert np.abs(r["keypoint_scores" + v] - g[f"p{i}_keypoint_scores{v}"].numpy()).max() < 5e-5
        counts.append((r["keypoints0"].shape[0], r["keypoints1"].shape[0]))
        pm = pairs_of(r["keypoints0"], r["keypoints1"], r["matches0"], r["matching_scores0"])
        pr = pairs_of(g[f"p{i}_keypoints0"].numpy(), g[f"p{i}_keypoints1"].numpy(), g[f"p{i}_matches0"].numpy(),
                      g[f"p{i}_matching_scores0"].numpy())
        assert len(pr) > 200 and set(pm) == set(pr), (name, len(pm), len(pr), len(set(pm) ^ set(pr)))
        assert max(abs(pm[q] - pr[q]) for q in pm) < 1e-4
        if counts[-1][0] < 1024 and counts[-1][1] < 1024:  # same key-point order as the reference: integers element-wise
            assert (r["matches0"] == g[f"p{i}_matches0"].numpy()).all() and (r["matches1"] == g[f"p{i}_matches1"].numpy()).all()
        total += len(pr)
        same += len(set(pm) & set(pr))
    assert any(a != b and max(a, b) < 1024 for a, b in counts), counts
    record(f"c3_official_ragged_pb{pair_batch}", ref_matches=total, identical=same, min_count=min(min(c) for c in counts))


def hpatches_shaped_list(n):
    """n loader items with the five image shapes of the fixtures in changing combinations (sequences of an
    HPatches-style list share a reference image: consecutive items repeat view 0's shape)."""
    shapes = [(480, 640), (480, 613), (640, 480), (725, 480), (480, 656)]
    items = []
    for i in range(n):
        s0, s1 = shapes[(i // 3) % 5], shapes[(i * 2 + 1) % 5]
        o0, o1 = (s0[1] * 2, s0[0] * 2), (int(s1[1] * 1.5), int(s1[0] * 1.5))
        items.append({"name": [f"s{i // 5}/{i % 5 + 2}.ppm"], **c3_pair(400 + i, s0, s1, (o0, o1))})
    return items


@pytest.mark.parametrize("pair_batch,workers", [(16, 1), (32, 1)])
def test_pair_batched_export_equals_sequential_loop(tmp_path, pair_batch, workers):
    """export_predictions(pair_batch=N): N consecutive pairs of DIFFERENT image shapes and key-point counts -- the
    extractor once per distinct shape, the matcher once over all N pairs (gfc_lg_forward_ragged) -- writes the records
    of the sequential batch-1 loop (utils/export_predictions.py:36-45): every integer output identical element-wise,
    floats within 1e-4.  40 HPatches-shaped pairs, five image shapes, detection threshold 0.5 so that views keep
    between ~850 and 1024 (the cap) key points."""
    items = hpatches_shaped_list(40)
    keys = EXPORT_KEYS + ["keypoint_scores0", "keypoint_scores1"]
    pipe = official_pipeline(C3_RAGGED_THRESHOLD)
    seq = load_predictions(export_predictions(items, pipe, tmp_path / "seq.npz", keys=keys))
    bat = load_predictions(export_predictions(items, pipe, tmp_path / "bat.npz", keys=keys, pair_batch=pair_batch,
                                              workers=workers))
    assert list(seq) == list(bat) and len(seq) == 40
    counts, ferr = set(), 0.0
    for name in seq:
        a, b = seq[name], bat[name]
        assert set(a) == set(b)
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (name, k)
            if a[k].dtype.kind in "iu":
                assert (a[k] == b[k]).all(), (name, k, int((a[k] != b[k]).sum()))
            else:
                ferr = max(ferr, float(np.abs(a[k] - b[k]).max()))
        counts.add(a["keypoints0"].shape[0])
        counts.add(a["keypoints1"].shape[0])
        assert (a["matches0"] >= 0).sum() > 100
    assert ferr < 1e-4, ferr
    assert len(counts) >= 5 and min(counts) < 1024 <= max(counts), sorted(counts)  # ragged, below and at the cap
    record(f"c3_pair_batch{pair_batch}_workers{workers}_vs_sequential", pairs=40, float_err=ferr,
           distinct_counts=len(counts), min_count=min(counts))


def _assert_records_equal(seq, other, n):
    assert list(seq) == list(other) and len(seq) == n
    ferr = 0.0
    for name in seq:
        a, b = seq[name], other[name]
        assert set(a) == set(b)
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (name, k)
            if a[k].dtype.kind in "iu":
                assert (a[k] == b[k]).all(), (name, k, int((a[k] != b[k]).sum()))
            else:
                ferr = max(ferr, float(np.abs(a[k] - b[k]).max()))
    assert ferr < 1e-4, ferr
    return ferr


@pytest.mark.parametrize("threshold", [0.0, C3_RAGGED_THRESHOLD])
def test_first_call_of_a_fresh_pipeline_is_the_pair_batched_export(tmp_path, threshold):
    """Round-4 finding: a NEVER-RUN pipeline whose first call is export_predictions(pair_batch=32) packs its extractor
    weights inside forward_views; the shape groups then run on two side streams.  The packing must be ordered before
    BOTH lanes (extract_views packs on the caller's stream before it forks).  72 pairs, five image shapes (>= 3 shape
    groups per batch), records against the sequential loop of ANOTHER never-run pipeline: integers identical
    element-wise, floats within 1e-4.  Likewise a never-run pipeline whose first call is `workers=2` (the workers share
    the weights the caller packed before their streams started)."""
    items = hpatches_shaped_list(72)
    keys = EXPORT_KEYS + ["keypoint_scores0", "keypoint_scores1"]
    seq = load_predictions(export_predictions(items, official_pipeline(threshold), tmp_path / "seq.npz", keys=keys))
    fresh = official_pipeline(threshold)
    assert fresh.extractor._packed is None and fresh.matcher.net._packed is None  # never run
    bat = load_predictions(export_predictions(items, fresh, tmp_path / "bat.npz", keys=keys, pair_batch=32))
    ferr = _assert_records_equal(seq, bat, 72)
    fresh2 = official_pipeline(threshold)
    par = load_predictions(export_predictions(items, fresh2, tmp_path / "par.npz", keys=keys, workers=2))
    ferr2 = _assert_records_equal(seq, par, 72)
    # and again on the now-warm pipelines (replicas / lanes are reused or rebuilt: same records)
    _assert_records_equal(seq, load_predictions(export_predictions(items, fresh, tmp_path / "bat2.npz", keys=keys,
                                                                   pair_batch=32)), 72)
    _assert_records_equal(seq, load_predictions(export_predictions(items, fresh2, tmp_path / "par2.npz", keys=keys,
                                                                   workers=3)), 72)
    total = sum(int((r["matches0"] >= 0).sum()) for r in seq.values())
    assert total > 72 * 100
    record(f"c3_fresh_pipeline_first_call_pb32_and_workers2_th{threshold}", pairs=72, matches_total=total,
           float_err_pair_batch=ferr, float_err_workers=ferr2)


def test_export_from_host_uint8_images_with_gpu_preprocessing(tmp_path):
    """The loop that FEEDS the path (datasets/hpatches.py:94-112 + utils/image.py:33-72 -> export_predictions.py:36-45):
    decoded uint8 images in pinned host memory -> asynchronous H2D copies -> gfc_preprocess_resize (short side 480,
    antialias) -> forward_pairs(pair_batch=32) -> records un-scaled by 1/scales, against the sequential loop fed by the
    CPU preprocessing of oracle/preprocess.py (numpy_image_to_torch + ImagePreprocessor restated; kornia's resize is
    absent offline: that half is unpinned) on the same bytes: identical key-point sets and matched coordinate pairs,
    scores within 1e-4; and against the SAME GPU-preprocessed images consumed pair by pair: integers identical."""
    from glue_factory_colon_amd import synthetic
    from glue_factory_colon_amd.image_preprocessor import HostImageFeeder
    from oracle import preprocess as opre

    raw = synthetic.hpatches_like_host_images(40, seed=5000)
    keys = EXPORT_KEYS + ["keypoint_scores0", "keypoint_scores1"]
    conf = {"resize": 480, "side": "short"}
    cpu_items = []
    for it in raw:
        item = {"name": [it["name"]]}
        for v in ("view0", "view1"):
            d = opre.preprocess(opre.numpy_image_to_torch(it[v]["image"].numpy()), resize=480, side="short")
            item[v] = {"image": d["image"][None], "scales": d["scales"][None],
                       "image_size": torch.from_numpy(d["image_size"])[None].float()}
        cpu_items.append(item)
    assert len({tuple(i[v]["image"].shape[-2:]) for i in cpu_items for v in ("view0", "view1")}) == 5
    seq = load_predictions(export_predictions(cpu_items, official_pipeline(), tmp_path / "seq.npz", keys=keys))
    feeder = HostImageFeeder(raw, conf)
    bat = load_predictions(export_predictions(feeder, official_pipeline(), tmp_path / "bat.npz", keys=keys, pair_batch=32))
    assert list(seq) == list(bat) and len(seq) == 40
    assert feeder.h2d_bytes == sum(it[v]["image"].numel() for it in raw for v in ("view0", "view1"))
    # The two loops see images that differ by the rounding of two resize implementations (<= 5e-7,
    # test_resize_gpu_vs_oracle): what may differ is the ORDER of two key points whose scores the extractor separates
    # by less than that (a near-tie swap inside the sorted top-k list, tests/parity_utils.py) -- then matches0/1 differ
    # as arrays while the matched coordinate pairs are the same set.  Demanded: identical key-point sets, identical
    # sets of matched coordinate pairs, scores within 1e-4; the number of pairs whose arrays are element-wise equal is
    # recorded.
    ferr = 0.0
    total = identical_arrays = 0
    for name in seq:
        a, b = seq[name], bat[name]
        same = True
        for k in a:
            assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, (name, k)
            if a[k].dtype.kind in "iu":
                same = same and bool((a[k] == b[k]).all())
        for v in "01":
            sa = {tuple(np.round(q, 2)) for q in a["keypoints" + v].tolist()}
            sb = {tuple(np.round(q, 2)) for q in b["keypoints" + v].tolist()}
            assert sa == sb, (name, v, len(sa ^ sb))
        pa = pairs_of(np.round(a["keypoints0"], 2), np.round(a["keypoints1"], 2), a["matches0"], a["matching_scores0"])
        pb = pairs_of(np.round(b["keypoints0"], 2), np.round(b["keypoints1"], 2), b["matches0"], b["matching_scores0"])
        assert set(pa) == set(pb), (name, len(pa), len(pb), len(set(pa) ^ set(pb)))
        ferr = max(ferr, max(abs(pa[q] - pb[q]) for q in pa))
        identical_arrays += int(same)
        total += len(pa)
    print(f"from-host vs CPU-preprocessed: score err {ferr:.3g}, {identical_arrays}/40 pairs with element-wise identical "
          f"arrays (the others: near-tie order swaps inside the top-k lists), {total} matches")
    # scores: the inputs of the two loops differ by <= 5e-7 (two resize implementations), which the matcher amplifies
    # (measured 2.5e-5); the 1e-4 bar of the path holds on IDENTICAL inputs (`one` below: bit-identical integers)
    assert ferr < 1e-4, ferr
    assert total > 40 * 100, total
    # the same feeder items consumed pair by pair (resident tensors, no batching): identical integers again
    one = load_predictions(export_predictions(list(HostImageFeeder(raw, conf)), official_pipeline(),
                                              tmp_path / "one.npz", keys=keys))
    _assert_records_equal(one, bat, 40)
    # the feeder consumed LIVE (not materialised into a list) by export workers on their own HIP streams: the items'
    # copies and resize kernels are queued on the caller's stream as the loader is iterated, the extractor runs on a
    # worker's stream -- ordered by an event per chunk and `record_stream` (export_predictions._export_loop); a small
    # `depth` makes the feeder recycle its device buffers while the workers are still reading earlier ones
    wk = load_predictions(export_predictions(HostImageFeeder(raw, conf, depth=2), official_pipeline(), tmp_path / "wk.npz",
                                             keys=keys, workers=2))
    _assert_records_equal(wk, bat, 40)
    # a float image is not a decoded image
    bad = [dict(raw[0]), {**raw[1], "view1": {"image": raw[1]["view1"]["image"].float()}}]
    with pytest.raises(ValueError, match="expected a decoded uint8 image"):
        list(HostImageFeeder(bad, conf))
    record("c3_from_host_uint8_pb32_vs_cpu_preprocessed_sequential", pairs=40, matches_total=total, float_err=ferr,
           pairs_with_elementwise_identical_arrays=identical_arrays,
           h2d_mb_per_pair=feeder.h2d_bytes / 40 / 1e6)


def test_view_dedupe_extracts_a_shared_reference_image_once_and_changes_no_record(tmp_path):
    """export_predictions(pair_batch=32, view_key=...): on an HPatches-structured list (every pair of a sequence carries
    the sequence's image 1 as view 0, datasets/hpatches.py:98-99) the views a key function names as the same image are
    extracted once per pair batch.  40 pairs = 8 sequences: 48 instead of 80 extractions; every record array is
    bit-identical to the run without keys (an image's features do not depend on the rest of its batch)."""
    from glue_factory_colon_amd import synthetic

    items = synthetic.hpatches_shaped_pairs(40, seed=6100, shared_view0=True)
    assert torch.equal(items[0]["view0"]["image"], items[4]["view0"]["image"])
    assert not torch.equal(items[0]["view0"]["image"], items[5]["view0"]["image"])
    keys = EXPORT_KEYS + ["keypoint_scores0", "keypoint_scores1"]
    counts = []

    def counting(pipe):
        inner = pipe.extractor.forward_views
        pipe.extractor.forward_views = lambda views: (counts.append(len(views)), inner(views))[1]
        return pipe

    plain = load_predictions(export_predictions(items, counting(official_pipeline()), tmp_path / "a.npz", keys=keys, pair_batch=32))
    n_plain = sum(counts)
    counts.clear()
    vk = lambda item, i: (item["scene"][0], 1) if i == 0 else None  # noqa: E731
    dedup = load_predictions(export_predictions(items, counting(official_pipeline()), tmp_path / "b.npz", keys=keys, pair_batch=32,
                                                view_key=vk))
    assert n_plain == 80 and sum(counts) == 40 + 7 + 2, (n_plain, counts)  # batch 1: 32 pairs = 7 sequences, batch 2: 8 pairs = 2
    assert list(plain) == list(dedup) and len(plain) == 40
    for name in plain:
        for k in plain[name]:
            assert np.array_equal(plain[name][k], dedup[name][k]), (name, k)
    assert sum(int((r["matches0"] >= 0).sum()) for r in plain.values()) > 40 * 100
    # names that collide on images of different shapes are refused
    bad = lambda item, i: "same"  # noqa: E731
    with pytest.raises(ValueError, match="different shapes"):
        export_predictions(items[:8], official_pipeline(), tmp_path / "c.npz", keys=keys, pair_batch=8, view_key=bad)


def test_workers_and_pair_batch_are_not_combinable(tmp_path):
    """Round-5 pruning: `workers` > 1 together with `pair_batch` > 1 is refused (slower than pair_batch alone)."""
    items = hpatches_shaped_list(4)
    with pytest.raises(ValueError, match="not combinable"):
        export_predictions(items, official_pipeline(), tmp_path / "x.npz", keys=EXPORT_KEYS, pair_batch=2, workers=2)


def test_forward_pairs_keys_and_single_pair_fallbacks():
    """TwoViewPipeline.forward_pairs returns, per pair, the keys of the single-pair call; one pair, cached features and
    pairs without key points in a view take the single-pair path."""
    pipe = official_pipeline(C3_RAGGED_THRESHOLD).to("cuda")
    datas = []
    for name, seed, s0, s1, origs in C3_RAGGED_PAIRS[:3]:
        d = c3_pair(seed, s0, s1, origs)
        datas.append({v: {k: t.to("cuda") for k, t in d[v].items()} for v in ("view0", "view1")})
    with torch.no_grad():
        single = [pipe(d) for d in datas]
        multi = pipe.forward_pairs(datas)
        assert len(pipe.forward_pairs(datas[:1])) == 1
        # a view without detections (threshold above every score): the reference's empty-set early return
        blank = official_pipeline(2.0).to("cuda")
        e = blank.forward_pairs(datas[:2])
    for a, b in zip(single, multi):
        assert set(a) == set(b)
        for k in ("matches0", "matches1"):
            assert torch.equal(a[k], b[k])
        for k in ("keypoints0", "keypoint_scores1", "descriptors0"):
            assert torch.equal(a[k], b[k])  # the extractor's per-image results do not depend on the batch
        assert (a["matching_scores0"] - b["matching_scores0"]).abs().max() < 1e-4
        assert a["log_assignment"].shape == b["log_assignment"].shape
        assert (a["log_assignment"] - b["log_assignment"]).abs().max() < 1e-4 * (1 + a["log_assignment"].abs().max())
        assert (a["ref_descriptors1"] - b["ref_descriptors1"]).abs().max() < 1e-4
    for p in e:
        assert p["keypoints0"].shape[1] == 0 and p["matches0"].shape == (1, 0) and p["log_assignment"].shape == (1, 1, 1)

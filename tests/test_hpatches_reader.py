"""The dependency-free reader of the files HPatches ships (glue_factory_colon_amd/hpatches.py): binary PPM / PGM decode,
`read_homography`, the pair list -- reference gluefactory/datasets/hpatches.py:23-35,38-77,94-112 and
gluefactory/utils/image.py:135-161.  A binary PPM is raw bytes behind a text header, so the files are written by the test
itself and the decode must return exactly those bytes."""
import os

import numpy as np
import pytest
import torch

from glue_factory_colon_amd import hpatches


def write_ppm(path, img, comment=False):
    h, w = img.shape[:2]
    magic = b"P6" if img.ndim == 3 else b"P5"
    head = magic + b"\n" + (b"# made by a test\n" if comment else b"") + f"{w} {h}\n255\n".encode()
    with open(path, "wb") as f:
        f.write(head + img.tobytes())


def make_tree(root, seqs, rng, size=(48, 64)):
    """<root>/<seq>/{1..6}.ppm and H_1_{2..6}; returns {seq: ([images], [H])}."""
    out = {}
    for s, seq in enumerate(seqs):
        d = root / seq
        d.mkdir(parents=True)
        imgs, hs = [], []
        for i in range(1, 7):
            h, w = size[0] + 2 * ((s + i) % 3), size[1] + 4 * (i % 2)
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            write_ppm(d / f"{i}.ppm", img, comment=i == 2)
            imgs.append(img)
        for q in range(2, 7):
            H = np.eye(3) + rng.normal(0, 0.01, (3, 3))
            H[2, 2] = 1.0
            hs.append(H)
            rows = ["   ".join(f"{v:.8g}" for v in row) + " " for row in H]  # runs of spaces, trailing space
            (d / f"H_1_{q}").write_text("\n".join(rows) + "\n\n")
        out[seq] = (imgs, hs)
    return out


def test_read_ppm_returns_the_bytes_of_the_file(tmp_path):
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (21, 34), dtype=np.uint8)
    write_ppm(tmp_path / "a.ppm", rgb, comment=True)
    write_ppm(tmp_path / "b.pgm", grey)
    a = hpatches.read_ppm(tmp_path / "a.ppm")
    assert a.dtype == np.uint8 and a.shape == (37, 53, 3) and np.array_equal(a, rgb)
    assert np.array_equal(hpatches.read_ppm(tmp_path / "b.pgm", grayscale=True), grey)
    b3 = hpatches.read_ppm(tmp_path / "b.pgm")  # a grey file read as colour: the channel three times
    assert b3.shape == (21, 34, 3) and all(np.array_equal(b3[..., c], grey) for c in range(3))
    g = hpatches.read_ppm(tmp_path / "a.ppm", grayscale=True)  # colour file read as grey: 0.299 R + 0.587 G + 0.114 B
    exact = rgb.astype(np.float64) @ np.array([0.299, 0.587, 0.114])
    assert g.shape == (37, 53) and np.abs(g.astype(np.float64) - exact).max() <= 0.51
    # the value 10 (a newline byte) as the first sample: exactly ONE whitespace byte ends the header
    tricky = np.full((2, 2, 3), 10, np.uint8)
    write_ppm(tmp_path / "t.ppm", tricky)
    assert np.array_equal(hpatches.read_ppm(tmp_path / "t.ppm"), tricky)
    # plain (ASCII) variant
    (tmp_path / "p.ppm").write_text("P3\n# c\n2 1\n255\n1 2 3\n250 251 252\n")
    assert hpatches.read_ppm(tmp_path / "p.ppm").tolist() == [[[1, 2, 3], [250, 251, 252]]]
    # the reference's errors (image.py:137-143)
    with pytest.raises(FileNotFoundError, match="No image at path"):
        hpatches.read_ppm(tmp_path / "missing.ppm")
    (tmp_path / "bad.ppm").write_bytes(b"\x89PNG....")
    with pytest.raises(IOError, match="Could not read image"):
        hpatches.read_ppm(tmp_path / "bad.ppm")
    (tmp_path / "short.ppm").write_bytes(b"P6\n4 4\n255\n" + bytes(10))
    with pytest.raises(IOError, match="Could not read image"):
        hpatches.read_ppm(tmp_path / "short.ppm")
    (tmp_path / "deep.ppm").write_bytes(b"P6\n1 1\n65535\n" + bytes(6))
    with pytest.raises(NotImplementedError):
        hpatches.read_ppm(tmp_path / "deep.ppm")


def test_read_homography_tolerates_the_formats_hpatches_uses(tmp_path):
    p = tmp_path / "H_1_2"
    p.write_text("0.87977   0.31245  -39.431 \n-0.18389  0.93847   153.16\n\n0.00019641 -1.6015e-05  1\n")
    H = hpatches.read_homography(p)
    assert H.dtype == np.float64 and H.shape == (3, 3)
    assert H.tolist() == [[0.87977, 0.31245, -39.431], [-0.18389, 0.93847, 153.16], [0.00019641, -1.6015e-05, 1.0]]


def test_pair_list_and_raw_items(tmp_path):
    rng = np.random.default_rng(1)
    seqs = ["i_ajuntament", "v_bark", "v_talent", "i_dc", "v_zzz"]
    tree = make_tree(tmp_path / "hp", seqs, rng)
    ds = hpatches.HPatches({"data_dir": str(tmp_path / "hp"), "preprocessing": {"resize": 32, "side": "short"},
                            "pin_memory": False})
    # sorted sequences, the large scenes of earlier papers dropped (hpatches.py:46-56,68-69), five pairs each
    assert ds.sequences == sorted(seqs) and len(ds) == 15
    assert [it[:2] for it in ds.items[:6]] == [("i_ajuntament", q) for q in range(2, 7)] + [("v_bark", 2)]
    assert len(hpatches.HPatches({"data_dir": str(tmp_path / "hp"), "ignore_large_images": False})) == 25
    assert {s for s, _, _ in hpatches.HPatches({"data_dir": str(tmp_path / "hp"), "subset": "v"}).items} == {"v_bark", "v_zzz"}
    it = ds[6]  # v_bark, q = 3
    imgs, hs = tree["v_bark"]
    assert it["name"] == "v_bark/6.ppm" and it["scene"] == "v_bark" and it["idx"].tolist() == [6]  # name as hpatches.py:108
    assert it["is_illu"].tolist() == [False] and ds[0]["is_illu"].tolist() == [True]
    assert torch.equal(it["view0"]["image"], torch.from_numpy(imgs[0])) and it["view0"]["image"].dtype == torch.uint8
    assert torch.equal(it["view1"]["image"], torch.from_numpy(imgs[2]))
    # H_0to1 = T1 . H . T0^-1 with T = diag(fp32(new_w / w), fp32(new_h / h), 1) (hpatches.py:102-103, image.py:49-50)
    pp = ds.preprocessor

    def T(img):
        h, w = img.shape[:2]
        nh, nw = pp.get_new_image_size(h, w)
        return np.diag([float(np.float32(nw / w)), float(np.float32(nh / h)), 1.0])

    Hfile = hpatches.read_homography(tmp_path / "hp" / "v_bark" / "H_1_3")
    assert np.abs(Hfile - hs[1]).max() < 1e-7
    want = (T(imgs[2]) @ Hfile @ np.linalg.inv(T(imgs[0]))).astype(np.float32)
    assert it["H_0to1"].shape == (1, 3, 3) and it["H_0to1"].dtype == torch.float32
    assert np.array_equal(it["H_0to1"][0].numpy(), want)
    assert len(list(iter(ds))) == 15
    # meta(): the same pair from the file headers alone (what the evaluation reads beside the cached predictions)
    m = ds.meta(6)
    assert m["name"] == it["name"] and m["scene"] == "v_bark" and torch.equal(m["H_0to1"], it["H_0to1"][0])
    for tag, img in (("view0", imgs[0]), ("view1", imgs[2])):
        h, w = img.shape[:2]
        nh, nw = pp.get_new_image_size(h, w)
        assert m[tag]["image_size"].tolist() == [nw, nh] and m[tag]["original_image_size"].tolist() == [w, h]
        assert torch.equal(m[tag]["scales"], torch.tensor([nw / w, nh / h], dtype=torch.float32))
    assert hpatches.HPatches.view_key(it, 0) == ("v_bark", 1) and hpatches.HPatches.view_key(it, 1) is None
    assert hpatches.HPatches.view_key({"scene": ["v_bark"]}, 0) == ("v_bark", 1)  # collated form
    # the same raw items from DataLoader worker processes (decode in parallel), whole list and a rank's share
    for indices in (None, [1, 6, 11]):
        got = list(ds.raw_loader(indices, num_workers=2))
        want = [ds[i] for i in (range(len(ds)) if indices is None else indices)]
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a["name"] == b["name"] and a["scene"] == b["scene"] and torch.equal(a["H_0to1"], b["H_0to1"])
            assert torch.equal(a["view0"]["image"], b["view0"]["image"]) and torch.equal(a["view1"]["image"], b["view1"]["image"])
    with pytest.raises(FileNotFoundError, match="no download"):
        hpatches.HPatches({"data_dir": str(tmp_path / "absent")})
    os.environ["GFC_DATA_PATH"] = str(tmp_path)
    try:
        assert len(hpatches.HPatches({"data_dir": "hp"})) == 15  # relative to $GFC_DATA_PATH
    finally:
        del os.environ["GFC_DATA_PATH"]


@pytest.mark.gpu
def test_export_from_an_hpatches_directory(tmp_path):
    """BASELINE config 3 from files: directory -> read_ppm -> HostImageFeeder -> export_predictions(pair_batch, view_key)
    -> records named like the reference's, identical to the records of the same decoded images handed over directly."""
    from glue_factory_colon_amd import synthetic
    from glue_factory_colon_amd.export_predictions import export_predictions, load_predictions
    from glue_factory_colon_amd.image_preprocessor import HostImageFeeder
    from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline

    raw = synthetic.hpatches_like_host_images(10, seed=5100, pin=False, shared_view0=True)  # two sequences of five pairs
    root = tmp_path / "hpatches-sequences-release"
    for i, it in enumerate(raw):
        seq = "v_" + it["scene"]
        (root / seq).mkdir(parents=True, exist_ok=True)
        if i % 5 == 0:
            write_ppm(root / seq / "1.ppm", it["view0"]["image"].numpy())
        write_ppm(root / seq / f"{i % 5 + 2}.ppm", it["view1"]["image"].numpy())
        (root / seq / f"H_1_{i % 5 + 2}").write_text("1 0 24\n0 1 16\n0 0 1\n")
    conf = {"resize": 480, "side": "short"}
    ds = hpatches.HPatches({"data_dir": str(root), "preprocessing": conf})
    assert len(ds) == 10

    def pipeline():
        return TwoViewPipeline({
            "extractor": {"name": "extractors.superpoint_open", "weights": "synthetic", "max_num_keypoints": 512,
                          "detection_threshold": 0.0, "nms_radius": 3},
            "matcher": {"name": "matchers.lightglue", "weights": "synthetic", "filter_threshold": 0.1}}).eval().cuda()

    keys = ["keypoints0", "keypoints1", "matches0", "matches1", "matching_scores0", "matching_scores1"]
    feeder = ds.feeder()
    from_files = load_predictions(export_predictions(feeder, pipeline(), tmp_path / "files.npz", keys=keys, pair_batch=8,
                                                     view_key=ds.view_key))
    assert sorted(from_files) == sorted(f"v_{it['scene']}/{i}.ppm" for i, it in enumerate(raw))
    # each sequence's image 1 travelled and was resized once
    assert feeder.h2d_bytes == sum(it["view1"]["image"].numel() for it in raw) + sum(raw[i]["view0"]["image"].numel() for i in (0, 5))
    # files decoded by two DataLoader worker processes, and a rank's share of the list (shard): the same records
    workers = ds.feeder(num_workers=2)
    par = load_predictions(export_predictions(workers, pipeline(), tmp_path / "workers.npz", keys=keys, pair_batch=8,
                                              view_key=ds.view_key))
    assert list(par) == list(from_files)
    for name, rec in par.items():
        for k in keys:
            assert np.array_equal(rec[k], from_files[name][k]), (name, k)
    share = list(ds.feeder(num_workers=2).shard(1, 2, group=5))
    assert [i for i, _ in share] == [5, 6, 7, 8, 9] and share[0][1]["name"] == [f"v_{raw[5]['scene']}/5.ppm"]
    direct_items = [{**it, "name": f"v_{it['scene']}/{i}.ppm"} for i, it in enumerate(raw)]
    direct = load_predictions(export_predictions(HostImageFeeder(direct_items, conf), pipeline(), tmp_path / "direct.npz",
                                                 keys=keys, pair_batch=8))
    for name, rec in direct.items():
        for k in keys:
            assert np.array_equal(rec[k], from_files[name][k]), (name, k)
    assert sum(int((r["matches0"] >= 0).sum()) for r in from_files.values()) > 10 * 50


@pytest.mark.gpu
def test_hpatches_pipeline_from_a_directory(tmp_path):
    """eval_hpatches.HPatchesPipeline (gluefactory/eval/hpatches.py:98-176 without the RANSAC estimators): directory ->
    predictions.h5 (reference layout) -> per-pair match metrics + DLT error on the GPU -> the reference's summary keys.
    The tree is two sequences whose pairs are crops of one canvas displaced by a known shift, so the written H_1_q is
    the true homography and the metrics have known-good values."""
    from glue_factory_colon_amd import _hdf5, eval_hpatches, synthetic
    from glue_factory_colon_amd.export_predictions import load_predictions

    raw = synthetic.hpatches_like_host_images(10, seed=5200, pin=False, shared_view0=True)
    root = tmp_path / "hpatches-sequences-release"
    shapes = {}
    for i, it in enumerate(raw):
        seq = "v_" + it["scene"]
        (root / seq).mkdir(parents=True, exist_ok=True)
        if i % 5 == 0:
            write_ppm(root / seq / "1.ppm", it["view0"]["image"].numpy())
        write_ppm(root / seq / f"{i % 5 + 2}.ppm", it["view1"]["image"].numpy())
        shapes[i] = (it["view0"]["image"].shape[:2], it["view1"]["image"].shape[:2])
    # view 1 = the canvas displaced by (dx, dy) = (24 + 6 k, 16 + 4 k) canvas pixels, k = i % 5, each view up-sampled from
    # its crop by its own factor: in ORIGINAL pixels x1 = (x0 / u0 - d) * u1 (synthetic.hpatches_like_host_images)
    from glue_factory_colon_amd.synthetic import HPATCHES_LIKE_ORIGINALS, HPATCHES_LIKE_SHAPES
    for i in range(10):
        j0, j1 = (i // 5) % 5, (i * 2 + 1) % 5
        u0x = HPATCHES_LIKE_ORIGINALS[j0][1] / HPATCHES_LIKE_SHAPES[j0][1]
        u0y = HPATCHES_LIKE_ORIGINALS[j0][0] / HPATCHES_LIKE_SHAPES[j0][0]
        u1x = HPATCHES_LIKE_ORIGINALS[j1][1] / HPATCHES_LIKE_SHAPES[j1][1]
        u1y = HPATCHES_LIKE_ORIGINALS[j1][0] / HPATCHES_LIKE_SHAPES[j1][0]
        dx, dy = 24 + 6 * (i % 5), 16 + 4 * (i % 5)
        H = np.array([[u1x / u0x, 0, -dx * u1x], [0, u1y / u0y, -dy * u1y], [0, 0, 1.0]])
        (root / ("v_" + raw[i]["scene"]) / f"H_1_{i % 5 + 2}").write_text("\n".join(" ".join(f"{v:.10g}" for v in row) for row in H) + "\n")
    pipe = eval_hpatches.HPatchesPipeline({"data_dir": str(root)}, pair_batch=8)
    model = eval_hpatches.build_model("synthetic", "synthetic", official=False, max_num_keypoints=512).cuda()
    summaries, results = pipe.run(tmp_path / "exp", model)
    pred_file = tmp_path / "exp" / "predictions.h5"
    assert pred_file.exists() and len(load_predictions(pred_file)) == 10
    if _hdf5.available():
        assert open(pred_file, "rb").read(4) == b"\x89HDF"  # the reference's container
    assert results["names"] == [f"v_{it['scene']}/{i}.ppm" for i, it in enumerate(raw)]
    for key in ("prec@1px", "prec@3px", "num_matches", "num_keypoints", "gt_match_recall@3px", "gt_match_precision@3px",
                "H_error_dlt"):
        assert f"mean_{key}" in summaries and f"med_{key}" in summaries, key
    for th in (1, 3, 5):
        assert f"H_error_dlt@{th}px" in summaries
    # The name-seeded matcher only finds the displacement its weights were calibrated on (k = 0: the first pair of each
    # sequence, ~340 matches; the larger shifts give ~10): there the matches follow the written homography -- precise at
    # 3 px, and the weighted DLT recovers it to well under a pixel.  The other pairs exercise the few-matches side.
    for i in (0, 5):
        assert results["num_matches"][i] > 200 and results["prec@3px"][i] > 0.95, (i, results["prec@3px"][i])
        assert results["H_error_dlt"][i] < 0.5, (i, results["H_error_dlt"][i])
    assert all(np.isfinite(results["prec@3px"])) and len(results["H_error_dlt"]) == 10
    assert summaries["mean_prec@3px"] == round(float(np.mean(results["prec@3px"])), 3)
    assert summaries["H_error_dlt@1px"] >= 0.19  # two of ten pairs below a pixel: AUC@1px >= 0.2 x (1 - err) (tools.py:137-149)
    # the grouped evaluation = the reference's pair-by-pair loop through CacheLoader and the drop-in eval functions
    pairwise = pipe.run_eval_pairwise(pred_file)
    assert pairwise["names"] == results["names"] and pairwise["num_matches"] == results["num_matches"]
    for key in (*eval_hpatches.eval_utils.RESULT_KEYS, "H_error_dlt"):
        a, b = np.array(pairwise[key], dtype=np.float64), np.array(results[key], dtype=np.float64)
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(b)) and np.allclose(a[fin], b[fin], rtol=1e-5, atol=1e-6), key
    # the command line (a child process: rendezvous-free single GPU run) prints the same summaries
    import json
    import subprocess
    import sys

    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "glue_factory_colon_amd.eval_hpatches", "--data_dir", str(root), "--open",
                        "--pair_batch", "8", "--experiment_dir", str(tmp_path / "cli")], capture_output=True, text=True,
                       cwd=root_dir, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    cli = json.loads(r.stdout[r.stdout.index("{"):])
    assert cli["mean_num_matches"] > 0 and set(cli) == set(summaries)
    # a second run reuses the prediction file (no model needed), as the reference does without --overwrite
    again, _ = pipe.run(tmp_path / "exp", None)
    assert again == summaries


def test_read_image_png_and_float_conversion(tmp_path, golden):
    """`read_image` / `load_image` on a PNG (the reference's assets are PNGs: assets/boat1.png; lossless, so the decode
    returns the encoder's pixels) and the float conversion of `numpy_image_to_torch` (image.py:135-161)."""
    PIL = pytest.importorskip("PIL.Image")
    from glue_factory_colon_amd import image_io

    rgb = golden("boat_native")["image0"].numpy()  # the reference's boat1.png as decoded uint8 RGB [680, 850, 3]
    PIL.fromarray(rgb).save(tmp_path / "boat1.png")
    back = image_io.read_image(tmp_path / "boat1.png")
    assert back.dtype == np.uint8 and np.array_equal(back, rgb)
    g = image_io.read_image(tmp_path / "boat1.png", grayscale=True)
    assert g.shape == rgb.shape[:2] and np.abs(g.astype(np.float64) - rgb.astype(np.float64) @ [0.299, 0.587, 0.114]).max() <= 0.51
    PIL.fromarray(rgb[..., 1]).save(tmp_path / "grey.png")
    assert np.array_equal(image_io.read_image(tmp_path / "grey.png", grayscale=True), rgb[..., 1])
    assert np.array_equal(image_io.read_image(tmp_path / "grey.png")[..., 2], rgb[..., 1])
    t = image_io.load_image(tmp_path / "boat1.png")
    assert t.dtype == torch.float32 and t.shape == (3, 680, 850)
    assert torch.equal(t, torch.tensor(rgb.transpose(2, 0, 1) / 255.0, dtype=torch.float))
    assert image_io.load_image(tmp_path / "grey.png", grayscale=True).shape == (1, 680, 850)
    (tmp_path / "junk.png").write_bytes(b"\x89PNG\r\n\x1a\n" + bytes(40))
    with pytest.raises(IOError, match="Could not read image"):
        image_io.read_image(tmp_path / "junk.png")

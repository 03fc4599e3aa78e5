"""'Next' row rank 1: image preprocessing.  The kornia resize is absent -> parity unpinned for the blur
parameters (see oracle/preprocess.py); the tests pin the oracle by properties and torch's own interpolate, then the
HIP kernel against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import preprocess as opp


def _img(h, w, c=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand((1, c, h // 8 + 2, w // 8 + 2), generator=g)
    img = F.interpolate(base, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)[0]
    return (img * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous()  # HWC uint8, like a decoded file


def test_oracle_size_logic_and_properties():
    # get_new_image_size: the HPatches setting (resize 480, side short) and the others (image.py:105-132)
    assert tuple(opp.get_new_image_size(600, 800, 480, "short")) == (480, 640)
    assert tuple(opp.get_new_image_size(800, 600, 480, "short")) == (640, 480)
    assert tuple(opp.get_new_image_size(600, 800, 480, "long")) == (360, 480)
    assert tuple(opp.get_new_image_size(600, 800, 480, "vert")) == (480, 640)
    assert tuple(opp.get_new_image_size(600, 800, 480, "horz")) == (360, 480)
    assert tuple(opp.get_new_image_size(600, 801, 481, "short", 8)) == (480, 640)
    assert tuple(opp.get_new_image_size(1, 1, [100, 120])) == (100, 120)
    u8 = _img(96, 128).numpy()
    x = opp.numpy_image_to_torch(u8)
    assert x.shape == (3, 96, 128) and x.dtype == torch.float32 and float(x.max()) <= 1.0
    # a constant image stays constant (normalised kernel, convex interpolation)
    c = torch.full((3, 90, 120), 0.3)
    assert torch.allclose(opp.kornia_resize(c, (45, 60)), torch.full((3, 45, 60), 0.3), atol=1e-6)
    # no blur when up-scaling or when antialias is off: exactly torch's interpolate
    up = opp.kornia_resize(x, (192, 256))
    assert torch.equal(up, F.interpolate(x[None], size=(192, 256), mode="bilinear", align_corners=None)[0])
    dn = opp.kornia_resize(x, (48, 64), antialias=False)
    assert torch.equal(dn, F.interpolate(x[None], size=(48, 64), mode="bilinear", align_corners=None)[0])
    # the blur parameters of the text: factor 2 -> sigma 0.5, 3 taps; factor 4 -> sigma 1.5, 7 taps
    k = opp.gaussian_kernel1d(3, 0.5)
    assert torch.allclose(k, torch.tensor([0.10650698, 0.78698604, 0.10650698]), atol=1e-6)
    d = opp.preprocess(x, resize=48, side="short")
    assert tuple(d["image"].shape) == (3, 48, 64) and list(d["image_size"]) == [64, 48]
    assert torch.allclose(d["scales"], torch.tensor([0.5, 0.5])) and list(d["original_image_size"]) == [128, 96]
    sq = opp.preprocess(x, resize=48, side="short", square_pad=True, add_padding_mask=True)
    assert tuple(sq["image"].shape) == (3, 64, 64) and int(sq["padding_mask"].sum()) == 48 * 64


@pytest.mark.gpu
@pytest.mark.parametrize("hw,size,ac,aa", [((96, 128), (48, 64), None, True), ((200, 301), (96, 144), None, True),
                                           ((240, 320), (60, 80), None, True), ((96, 128), (192, 256), None, True),
                                           ((97, 131), (50, 77), True, True), ((97, 131), (50, 77), None, False),
                                           ((300, 100), (40, 90), None, True)])
def test_resize_gpu_vs_oracle(hw, size, ac, aa):
    from glue_factory_colon_amd import image_preprocessor as ip

    u8 = _img(hw[0], hw[1], 3, seed=hw[0])
    x = opp.numpy_image_to_torch(u8.numpy())
    ref = opp.kornia_resize(x, size, ac, aa)
    out_f = ip.resize(x.cuda(), size, ac, aa).cpu()
    out_u = ip.resize(u8.cuda(), size, ac, aa).cpu()
    assert torch.equal(out_f, out_u)  # the fused uint8 path converts exactly like numpy_image_to_torch
    assert out_f.shape == ref.shape
    assert float((out_f - ref).abs().max()) <= 5e-7, float((out_f - ref).abs().max())
    # grey image, batched float input, BGR flip
    g8 = u8[..., 0].contiguous()
    og = ip.resize(g8.cuda(), size, ac, aa).cpu()
    assert float((og - ref[:1]).abs().max()) <= 5e-7
    ob = ip.resize(torch.stack([x, x.flip(0)]).cuda(), size, ac, aa).cpu()
    assert torch.equal(ob[0], out_f) and torch.equal(ob[1], out_f.flip(0))
    obgr = ip.resize(u8.flip(-1).contiguous().cuda(), size, ac, aa, bgr=True).cpu()
    assert torch.equal(obgr, out_f)


@pytest.mark.gpu
def test_image_preprocessor_contract():
    from glue_factory_colon_amd.image_preprocessor import ImagePreprocessor

    u8 = _img(120, 160, 3, seed=5)
    x = opp.numpy_image_to_torch(u8.numpy())
    pp = ImagePreprocessor({"resize": 60, "side": "short"})
    for inp in (x.cuda(), u8.cuda()):
        d = pp(inp)
        ref = opp.preprocess(x, resize=60, side="short")
        assert sorted(d) == sorted(ref)
        assert float((d["image"].cpu() - ref["image"]).abs().max()) <= 5e-7
        assert torch.allclose(d["scales"].cpu(), ref["scales"]) and list(d["image_size"]) == list(ref["image_size"])
        assert np.allclose(d["transform"], ref["transform"]) and list(d["original_image_size"]) == [160, 120]
    sq = ImagePreprocessor({"resize": 60, "side": "short", "square_pad": True, "add_padding_mask": True})(x.cuda())
    assert tuple(sq["image"].shape) == (3, 80, 80) and int(sq["padding_mask"].sum()) == 60 * 80
    same = ImagePreprocessor({})(x.cuda())
    assert torch.equal(same["image"].cpu(), x) and list(same["image_size"]) == [160, 120]
    with pytest.raises(KeyError):
        ImagePreprocessor({"resise": 3})
    with pytest.raises(NotImplementedError):
        ImagePreprocessor({"resize": 60, "interpolation": "bicubic"})(x.cuda())
    # extractor consumes the preprocessed view directly
    from glue_factory_colon_amd import superpoint_open
    ext = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": 128}).eval().cuda()
    big = _img(240, 328, 3, seed=9)
    view = ImagePreprocessor({"resize": 120, "side": "short", "edge_divisible_by": 8})(big.cuda())
    p = ext({"image": view["image"][None]})
    assert p["keypoints"].shape[1] == 128 and float(p["keypoints"][..., 0].max()) < view["image_size"][0]


@pytest.mark.gpu
def test_host_image_feeder_items_and_round_robin_shard():
    """HostImageFeeder: decoded uint8 host images (RGB and grey) -> loader items on the GPU with the fields the reference's
    HPatches loader collates (datasets/hpatches.py:94-112: image, scales, image_size, original_image_size, transform);
    `shard(rank, world)` yields this rank's round-robin share and touches only its own images."""
    from glue_factory_colon_amd.image_preprocessor import HostImageFeeder, ImagePreprocessor

    raw = []
    for i in range(7):
        h, w = (90 + 7 * i, 120 + 5 * i)
        img = _img(h, w, 3, seed=20 + i)
        raw.append({"name": f"s/{i}.ppm", "idx": i, "view0": {"image": img.pin_memory()},
                    "view1": {"image": img[..., 0].contiguous()}})  # view 1: a grey image [H,W]
    conf = {"resize": 64, "side": "short"}
    feeder = HostImageFeeder(raw, conf, depth=3)
    items = list(feeder)
    assert len(items) == 7 and feeder.h2d_bytes == sum(r[v]["image"].numel() for r in raw for v in ("view0", "view1"))
    pp = ImagePreprocessor(conf)
    for r, it in zip(raw, items):
        assert it["name"] == [r["name"]] and it["idx"] == r["idx"]
        for v, c in (("view0", 3), ("view1", 1)):
            ref = pp(r[v]["image"].cuda())
            d = it[v]
            assert d["image"].shape[:2] == (1, c) and torch.equal(d["image"][0], ref["image"])
            assert torch.equal(d["scales"][0].cpu(), ref["scales"].cpu())
            assert d["image_size"][0].tolist() == list(ref["image_size"])
            assert d["original_image_size"][0].tolist() == list(ref["original_image_size"])
            assert np.allclose(d["transform"][0].numpy(), ref["transform"])
    share = list(HostImageFeeder(raw, conf, depth=2).shard(1, 3))
    assert [i for i, _ in share] == [1, 4]
    for i, it in share:
        assert torch.equal(it["view0"]["image"], items[i]["view0"]["image"]) and it["name"] == items[i]["name"]
    # the reference's image_size / original_image_size are integer HOST tensors after collation; here: the same integers
    # as float32 DEVICE tensors (documented difference, HostImageFeeder docstring)
    d = items[0]["view0"]
    assert d["image_size"].dtype == torch.float32 and d["image_size"].is_cuda and d["image_size"].shape == (1, 2)
    assert torch.equal(d["image_size"], d["image_size"].round()) and torch.equal(d["original_image_size"][0].cpu(),
                                                                                 torch.tensor([120.0, 90.0]))
    # raw items from a plain iterable (no len(), no indexing: e.g. a DataLoader with batch_size=None): shard() walks it and
    # stages only this rank's items; grouped round-robin; two live iterators over ONE feeder do not disturb each other
    lazy = HostImageFeeder((r for r in raw), conf, depth=2)
    with pytest.raises(TypeError):
        len(lazy)
    share = list(lazy.shard(1, 2, group=2))
    assert [i for i, _ in share] == [2, 3, 6] and lazy.h2d_bytes == sum(raw[i][v]["image"].numel() for i in (2, 3, 6)
                                                                        for v in ("view0", "view1"))
    for i, it in share:
        assert torch.equal(it["view1"]["image"], items[i]["view1"]["image"])
    both = HostImageFeeder(raw, conf, depth=2)
    a, b = iter(both), both.shard(0, 2)
    mixed = [next(a), next(b)[1], next(a), next(b)[1], next(a)]
    for it, i in zip(mixed, (0, 0, 1, 2, 2)):
        assert torch.equal(it["view0"]["image"], items[i]["view0"]["image"]) and torch.equal(it["view0"]["scales"], items[i]["view0"]["scales"])


@pytest.mark.gpu
def test_host_image_feeder_copies_and_resizes_a_named_image_once():
    """HostImageFeeder(view_key=...): an image named again within the last `keep` names is neither copied nor resized a second
    time -- its items share the first occurrence's tensors; the items equal those of the feeder without names."""
    from glue_factory_colon_amd.image_preprocessor import HostImageFeeder

    ref_img = [_img(96, 128, 3, seed=70 + s) for s in range(3)]
    raw = [{"name": f"s{i // 3}/{i % 3 + 2}.ppm", "scene": f"s{i // 3}", "view0": {"image": ref_img[i // 3]},
            "view1": {"image": _img(90 + i, 120, 3, seed=80 + i)}} for i in range(9)]
    conf = {"resize": 64, "side": "short"}
    plain = list(HostImageFeeder(raw, conf, depth=4))
    feeder = HostImageFeeder(raw, conf, depth=4, view_key=lambda r, i: (r["scene"], 1) if i == 0 else None)
    named = list(feeder)
    total = sum(r[v]["image"].numel() for r in raw for v in ("view0", "view1"))
    assert feeder.h2d_bytes == total - 6 * ref_img[0].numel()  # two of the three occurrences of each reference image skipped
    for i, (a, b) in enumerate(zip(plain, named)):
        assert a["name"] == b["name"] == [raw[i]["name"]] and b["scene"] == [raw[i]["scene"]]
        for v in ("view0", "view1"):
            for k in ("image", "scales", "image_size", "original_image_size"):
                assert torch.equal(a[v][k], b[v][k]), (i, v, k)
    assert named[0]["view0"]["image"].data_ptr() == named[2]["view0"]["image"].data_ptr()
    assert named[0]["view0"]["image"].data_ptr() != named[3]["view0"]["image"].data_ptr()
    # `keep` bounds the window: with keep = 1 a name is forgotten once another named image has passed
    f1 = HostImageFeeder(raw[:4], conf, view_key=lambda r, i: (r["scene"], 0) if i == 0 else r["name"], keep=1)
    list(f1)
    assert f1.h2d_bytes == sum(r[v]["image"].numel() for r in raw[:4] for v in ("view0", "view1"))

"""DISK extractor row (BASELINE config 5): the stages behind the network as HIP kernels
(csrc/disk_detect.hip) and the boundary module (glue_factory_colon_amd.disk_kornia) against oracle/disk.py, the
restatement of gluefactory/models/extractors/disk_kornia.py:29-137 + the kornia functions it calls (kornia is absent:
parity for those is UNPINNED, see the oracle's header).  Index outputs bit-exact, descriptors within 1e-6."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from glue_factory_colon_amd import _native as nat  # noqa: E402
from glue_factory_colon_amd import disk_kornia, lightglue_pretrained  # noqa: E402
from glue_factory_colon_amd.registry import get_model  # noqa: E402
from oracle import disk as odisk  # noqa: E402

DEV = "cuda"


def heatmaps(b, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    s = torch.randn((b, 1, h, w), generator=g)
    s[0] = (s[0] * 4).round() / 4            # plateaus / exact ties: the first maximum of a window wins
    if b > 1:
        s[1, 0, 10:14, 5:25] = 2.5           # constant patch
    if b > 2:
        s[2] = -1.0                          # nothing above the cutoff ...
        s[2, 0, 7, 9] = 0.3                  # ... but one pixel
    return s


@pytest.mark.parametrize("h,w,n,window,cutoff", [(48, 64, 50, 5, 0.0), (37, 53, 2000, 5, 0.0), (40, 40, None, 5, 0.0),
                                                 (64, 96, 100, 3, 0.5), (33, 70, 7, 7, -10.0), (20, 20, 30, 1, 0.0)])
def test_disk_nms_select_vs_oracle(h, w, n, window, cutoff):
    lib = nat.lib()
    heat = heatmaps(3, h, w, h * w + (n or 0))
    ref = odisk.heatmap_to_keypoints(heat, n, window, cutoff)
    cap = n if n is not None else h * w
    kp = torch.full((3, cap, 2), -7.0, device=DEV)
    sc = torch.full((3, cap), -7.0, device=DEV)
    cnt = torch.empty((3,), dtype=torch.int32, device=DEV)
    ws = torch.empty(lib.gfc_disk_select_workspace_bytes(3, h, w), dtype=torch.uint8, device=DEV)
    hd = heat.reshape(3, h, w).contiguous().to(DEV)
    nat.check(lib.gfc_disk_nms_select(nat.ptr(hd), 3, h, w, window, cutoff, -1 if n is None else n, cap, nat.ptr(kp),
                                      nat.ptr(sc), nat.ptr(cnt), nat.ptr(ws), ws.numel(),
                                      nat.stream_ptr(torch.device(DEV))), "disk_nms_select")
    torch.cuda.synchronize()
    for i, (xy, s) in enumerate(ref):
        c = int(cnt[i])
        assert c == xy.shape[0], (i, c, xy.shape)
        assert torch.equal(kp[i, :c].cpu(), xy.float()) and torch.equal(sc[i, :c].cpu(), s)
    if cutoff >= 0:  # image 2 has ONE candidate: with n set, kornia's (n+1)-th-score threshold drops it
        assert int(cnt[2]) == (0 if n is not None else 1)
    assert lib.gfc_disk_nms_select(nat.ptr(hd), 3, h, w, 4, 0.0, 5, 5, nat.ptr(kp), nat.ptr(sc), nat.ptr(cnt),
                                   nat.ptr(ws), ws.numel(), None) == 1  # even window: invalid, as kornia raises


def test_disk_gather_descriptors_vs_oracle():
    lib = nat.lib()
    g = torch.Generator().manual_seed(3)
    b, d, h, w, cap = 2, 128, 30, 41, 60
    dense = torch.randn((b, d, h, w), generator=g)
    dense[0, :, 3, 4] = 0.0  # zero vector: F.normalize's eps path
    xy = torch.stack([torch.randint(0, w, (b, cap), generator=g), torch.randint(0, h, (b, cap), generator=g)], -1)
    xy[0, 0] = torch.tensor([4, 3])
    counts = torch.tensor([cap, 17], dtype=torch.int32)
    out = torch.full((b, cap, d), float("nan"), device=DEV)
    dd, xd, cd = dense.to(DEV), xy.float().to(DEV), counts.to(DEV)  # kept alive: the library gets raw pointers
    nat.check(lib.gfc_disk_gather_descriptors(nat.ptr(dd), b, d, h, w, nat.ptr(xd), nat.ptr(cd), cap, nat.ptr(out),
                                              nat.stream_ptr(torch.device(DEV))), "gather")
    torch.cuda.synchronize()
    for i in range(b):
        c = int(counts[i])
        ref = odisk.merge_with_descriptors(xy[i, :c], dense[i])
        assert (out[i, :c].cpu() - ref).abs().max() < 1e-6
        assert (out[i, c:] == 0).all()


def fake_dense(images):
    """A stand-in for the absent network (plumbing for this test only, torch ops on the device): 1-channel heat-map
    and 128-channel descriptors from fixed random 3x3 filters.  Input sides are multiples of 16, as DISK's U-Net needs."""
    assert images.shape[-1] % 16 == 0 and images.shape[-2] % 16 == 0 and images.shape[1] == 3
    g = torch.Generator().manual_seed(99)
    wt = (torch.randn((129, 3, 3, 3), generator=g) * 0.5).to(images)
    y = F.conv2d(images, wt, padding=1)
    return y[:, 128:] * 3 - 0.2, y[:, :128]


@pytest.mark.parametrize("b,h,w,k", [(1, 64, 96, 80), (6, 50, 71, 40), (1, 33, 47, None)])
def test_disk_module_vs_oracle(b, h, w, k):
    """Module: pad to /16, chunks of 4, crop, NMS + top-n, descriptors, +0.5; with and without specular mask."""
    assert get_model("extractors.disk_kornia") is disk_kornia.DISK
    g = torch.Generator().manual_seed(b * 100 + h)
    img = torch.rand((b, 3, h, w), generator=g)
    m = disk_kornia.DISK({"max_num_keypoints": k, "force_num_keypoints": b > 1 and k is not None},
                         dense_fn=fake_dense).eval()
    assert m.is_initialized()
    pred = m({"image": img.to(DEV)})
    cpu_dense = lambda x: tuple(t.cpu() for t in fake_dense(x.to(DEV)))  # noqa: E731  same arithmetic as the module saw
    kps, scs, des = odisk.extract(cpu_dense, img, max_num_keypoints=k)
    for i in range(b):
        c = kps[i].shape[0]
        if b == 1 or k is None:
            if b > 1 and len({x.shape[0] for x in kps}) > 1:
                continue
            assert pred["keypoints"].shape[1] == c
        assert torch.equal(pred["keypoints"][i, :c].cpu(), kps[i]), i
        assert torch.equal(pred["keypoint_scores"][i, :c].cpu(), scs[i])
        assert (pred["descriptors"][i, :c].cpu() - des[i]).abs().max() < 1e-6
        if k is not None and b > 1:  # force_num_keypoints: zero scores / descriptors behind the real points
            assert pred["keypoints"].shape == (b, k, 2) and (pred["keypoint_scores"][i, c:] == 0).all()
            assert (pred["descriptors"][i, c:] == 0).all()
    assert pred["descriptors"].shape[-1] == 128 and pred["extractor_core_time_ms"].shape == (b,)
    # specular mask (disk_kornia.py:84-107), cropped by image_size
    mask = torch.rand((b, 1, h, w), generator=g) > 0.3
    size = torch.tensor([[w - 3.0, h - 2.0]] * b)
    mf = disk_kornia.DISK({"max_num_keypoints": k, "force_num_keypoints": True if k is not None else False},
                          dense_fn=fake_dense).eval()
    if k is not None:
        pred = mf({"image": img.to(DEV), "specular_mask": mask.to(DEV), "image_size": size.to(DEV)})
        kps, scs, des = odisk.extract(cpu_dense, img, max_num_keypoints=k, specular_mask=mask, image_size=size)
        for i in range(b):
            c = kps[i].shape[0]
            assert torch.equal(pred["keypoints"][i, :c].cpu(), kps[i]) and (pred["keypoint_scores"][i, c:] == 0).all()
            assert (pred["descriptors"][i, :c].cpu() - des[i]).abs().max() < 1e-6


def test_disk_into_lightglue_128d_and_errors():
    """Config 5's API surface: DISK-shaped features (128-d) feed the LightGlue side built for them."""
    img = torch.rand((1, 3, 64, 80), generator=torch.Generator().manual_seed(1)).to(DEV)
    ext = disk_kornia.DISK({"max_num_keypoints": 64}, dense_fn=fake_dense).eval()
    p0, p1 = ext({"image": img}), ext({"image": img.flip(-1).contiguous()})
    mat = lightglue_pretrained.LightGlue({"features": "disk", "weights": "synthetic"}).eval().to(DEV)
    size = torch.tensor([[80.0, 64.0]], device=DEV)
    out = mat({"keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"], "descriptors0": p0["descriptors"],
               "descriptors1": p1["descriptors"], "view0": {"image_size": size}, "view1": {"image_size": size}})
    assert out["matches0"].shape == (1, p0["keypoints"].shape[1]) and out["matches0"].dtype == torch.int64
    with pytest.raises(AssertionError, match="Missing key image"):
        ext({"img": img})
    with pytest.raises(RuntimeError, match="different numbers of keypoints"):  # the reference's torch.stack raises too
        two = torch.cat([img, img * 0.0], 0)
        disk_kornia.DISK({"max_num_keypoints": None}, dense_fn=fake_dense).eval()({"image": two})
    bare = disk_kornia.DISK({"max_num_keypoints": 64}).eval()  # weights "depth": a download in kornia, nothing here
    assert not bare.is_initialized()
    with pytest.raises(RuntimeError, match="has no weights"):
        bare({"image": img})

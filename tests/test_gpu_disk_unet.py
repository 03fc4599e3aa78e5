"""DISK's network (BASELINE config 5) on the HIP kernels of csrc/disk_unet.hip against oracle/disk_unet.py,
the restatement of kornia's thin U-Net (kornia is absent: NETWORK PARITY UNPINNED, see the oracle's header).
Every kernel against torch float64 first, then the whole network and the extractor module."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from glue_factory_colon_amd import _native as nat  # noqa: E402
from glue_factory_colon_amd import disk_kornia, weights  # noqa: E402
from glue_factory_colon_amd.disk_unet import DiskUnet  # noqa: E402
from oracle import disk as odisk  # noqa: E402
from oracle import disk_unet as ounet  # noqa: E402

DEV = torch.device("cuda", 0)


def st():
    return nat.stream_ptr(DEV)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


@pytest.mark.parametrize("cin,cout,h,w,gated,b", [(4, 16, 32, 48, False, 2), (16, 32, 33, 21, True, 2), (64, 64, 16, 16, True, 1),
                                                  (128, 64, 30, 40, True, 2), (96, 64, 17, 50, True, 1), (80, 129, 48, 35, True, 2),
                                                  (32, 100, 20, 20, True, 1), (16, 96, 5, 7, False, 3)])
def test_conv5x5_vs_torch(cin, cout, h, w, gated, b):
    """[InstanceNorm2d -> PReLU ->] Conv2d(5x5, pad 2) into a channel slice of a wider tensor, every output-channel split
    of the launcher (64-wide blocks, > 32 remainder, <= 32 remainder) and ragged tiles."""
    lib = nat.lib()
    g = torch.Generator().manual_seed(cin * 1000 + cout + h)
    x = torch.randn((b, cin, h, w), generator=g) * 2 + 0.5
    wt = torch.randn((cout, cin, 5, 5), generator=g) / (5 * cin ** 0.5)
    bias = torch.randn((cout,), generator=g) * 0.1
    slope = torch.rand((cin,), generator=g) * 0.5
    ref = x.double()
    if gated:
        ref = F.prelu(F.instance_norm(ref, eps=1e-5), slope.double())
    ref = F.conv2d(ref, wt.double(), bias.double(), padding=2)
    xd = nhwc(x)
    wp = torch.empty((lib.gfc_disk_conv5x5_packed_floats(cout, cin),), device=DEV)
    nat.check(lib.gfc_disk_pack_conv5x5(nat.ptr(wt.to(DEV)), nat.ptr(wp), cout, cin, st()), "pack")
    mean = rstd = sl = None
    if gated:
        mean, rstd, sl = torch.empty((b, cin), device=DEV), torch.empty((b, cin), device=DEV), slope.to(DEV)
        ws = torch.empty(lib.gfc_disk_instnorm_workspace_bytes(b, cin), dtype=torch.uint8, device=DEV)
        nat.check(lib.gfc_disk_instnorm_stats(nat.ptr(xd), b, h, w, cin, 1e-5, nat.ptr(mean), nat.ptr(rstd), nat.ptr(ws),
                                              ws.numel(), st()), "stats")
        xr = x.double()
        assert (mean.cpu().double() - xr.mean((2, 3))).abs().max() < 1e-6
        assert (rstd.cpu().double() * (xr.var((2, 3), unbiased=False) + 1e-5).sqrt() - 1).abs().max() < 1e-6
    off, ld = 8, cout + 8 + 4  # a slice [8, 8 + cout) of a wider tensor; the rest must stay untouched
    y = torch.full((b, h, w, ld), float("nan"), device=DEV)
    ys = y[..., off:]
    bd = bias.to(DEV)
    nat.check(lib.gfc_disk_conv5x5(nat.ptr(xd), nat.ptr(mean), nat.ptr(rstd), nat.ptr(sl), nat.ptr(wp), nat.ptr(bd),
                                   ys.data_ptr(), ld, b, h, w, cin, cout, 0, cout, st()), "conv5x5")
    torch.cuda.synchronize()
    got = y[..., off:off + cout].permute(0, 3, 1, 2).double().cpu()
    err = ((got - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 2e-5, err
    assert torch.isnan(y[..., :off]).all() and torch.isnan(y[..., off + cout:]).all()
    if cout > 32:  # a channel range of the layer (the last layer's descriptors | heat-map split)
        first = 32 * ((cout - 1) // 32)
        y2 = torch.full((b, h, w, cout - first), float("nan"), device=DEV)
        nat.check(lib.gfc_disk_conv5x5(nat.ptr(xd), nat.ptr(mean), nat.ptr(rstd), nat.ptr(sl), nat.ptr(wp), nat.ptr(bd),
                                       nat.ptr(y2), cout - first, b, h, w, cin, cout, first, cout - first, st()), "conv5x5")
        torch.cuda.synchronize()
        assert torch.equal(y2, y[..., off + first:off + cout].contiguous())
    from parity_utils import record
    record(f"disk_conv5x5_{cin}_{cout}_{h}x{w}", rel_err=err)


def test_conv5x5_rejects_bad_arguments():
    lib = nat.lib()
    x = torch.zeros(4096, device=DEV)
    args = lambda **k: [k.get(n, d) for n, d in (("cin", 16), ("cout", 32), ("first", 0), ("count", 32))]  # noqa: E731

    def call(ldy=32, **k):
        cin, cout, first, count = args(**k)
        return lib.gfc_disk_conv5x5(nat.ptr(x), None, None, None, nat.ptr(x), None, nat.ptr(x), ldy, 1, 4, 4, cin, cout, first,
                                    count, st())
    assert call(cin=6) == 1 and call(first=16) == 1 and call(count=40) == 1 and call(ldy=8) == 1
    assert lib.gfc_disk_conv5x5(nat.ptr(x), nat.ptr(x), None, None, nat.ptr(x), None, nat.ptr(x), 32, 1, 4, 4, 16, 32, 0, 32, st()) == 1
    assert lib.gfc_disk_instnorm_stats(nat.ptr(x), 1, 4, 4, 28, 1e-5, nat.ptr(x), nat.ptr(x), nat.ptr(x), 1 << 20, st()) == 3  # 480 % 7 != 0: unsupported
    assert lib.gfc_disk_avgpool2(nat.ptr(x), 16, 1, 5, 4, 16, nat.ptr(x), st()) == 1  # odd height


@pytest.mark.parametrize("c,h,w,b", [(16, 32, 48, 2), (64, 6, 10, 3), (32, 2, 2, 1)])
def test_pool_upsample_layout_vs_torch(c, h, w, b):
    lib = nat.lib()
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn((b, c, h, w), generator=g)
    wide = torch.randn((b, h, w, c + 12), generator=g).to(DEV)  # the pooled tensor is a channel slice of a wider one
    wide[..., 4:4 + c] = nhwc(x)
    y = torch.empty((b, h // 2, w // 2, c), device=DEV)
    nat.check(lib.gfc_disk_avgpool2(wide[..., 4:].data_ptr(), c + 12, b, h, w, c, nat.ptr(y), st()), "avgpool")
    assert (y.permute(0, 3, 1, 2).cpu() - F.avg_pool2d(x, 2)).abs().max() < 1e-6
    up = torch.full((b, 2 * h, 2 * w, c + 8), float("nan"), device=DEV)
    nat.check(lib.gfc_disk_upsample2(nat.ptr(nhwc(x)), b, h, w, c, nat.ptr(up), c + 8, st()), "upsample")
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    assert (up[..., :c].permute(0, 3, 1, 2).cpu() - ref).abs().max() < 1e-6 and torch.isnan(up[..., c:]).all()
    img = torch.rand((b, 3, h, w), generator=g)
    x4 = torch.empty((b, h, w, 4), device=DEV)
    nat.check(lib.gfc_disk_nchw3_to_nhwc4(nat.ptr(img.to(DEV)), b, h, w, nat.ptr(x4), st()), "nhwc4")
    assert torch.equal(x4[..., :3].cpu(), img.permute(0, 2, 3, 1)) and (x4[..., 3] == 0).all()


@pytest.mark.parametrize("b,h,w", [(2, 64, 96), (1, 480, 640), (3, 32, 16)])
def test_unet_vs_oracle(b, h, w):
    """The whole network (9 convolutions, 8 statistics passes, 4 poolings, 4 up-samplings) against the CPU restatement,
    same name-seeded weights through kornia's key names."""
    sd = weights.disk_state_dict(0)
    net = DiskUnet(128).to(DEV)
    net.load_state_dict(sd)
    assert set(net.state_dict()) == set(sd)
    img = torch.rand((b, 3, h, w), generator=torch.Generator().manual_seed(h + b))
    heat, desc = net.heatmap_and_dense_descriptors(img.to(DEV))
    rh, rd = ounet.heatmap_and_dense_descriptors(sd, img)
    assert heat.shape == rh.shape and desc.shape == rd.shape
    eh = ((heat.cpu() - rh).abs() / (1 + rh.abs())).max().item()
    ed = ((desc.cpu() - rd).abs() / (1 + rd.abs())).max().item()
    from parity_utils import record
    record(f"disk_unet_{b}x{h}x{w}", heat_rel_err=eh, desc_rel_err=ed, heat_std=rh.std().item())
    assert eh < 1e-4 and ed < 1e-4, (eh, ed)
    with pytest.raises(ValueError, match="divisible by 16"):
        net.dense_nhwc(torch.zeros((1, 3, 40, 64), device=DEV))
    with pytest.raises(ValueError, match="more than 1 spatial element"):  # as torch's instance_norm at the 1 x 1 level
        net.dense_nhwc(torch.zeros((1, 3, 16, 16), device=DEV))


@pytest.mark.parametrize("b,h,w,k", [(1, 64, 96, 80), (2, 50, 71, 40), (1, 480, 640, 2048)])
def test_disk_module_native_network(b, h, w, k):
    """The extractor module on its own network (`weights: synthetic`): pad to /16, network, crop, NMS + top-n, descriptors.
    (a) the stages behind the network bit-exact against oracle/disk.py fed with the module's own dense outputs;
    (b) the whole path against the CPU restatement end to end: key points that differ must be near-ties of the heat-map."""
    img = torch.rand((b, 3, h, w), generator=torch.Generator().manual_seed(b * 100 + h))
    m = disk_kornia.DISK({"max_num_keypoints": k, "force_num_keypoints": b > 1, "weights": "synthetic",
                          "dense_outputs": True}).eval().to(DEV)
    assert m.is_initialized() and all(n.startswith("model.unet.") for n in m.state_dict())
    pred = m({"image": img.to(DEV)})
    assert pred["dense_descriptors"].shape == (b, 128, h, w)

    def own_dense(x):  # the network as the module ran it (padded input -> NCHW views), on the CPU for the oracle
        hm, ds = m.model.heatmap_and_dense_descriptors(x.to(DEV))
        return hm.cpu(), ds.cpu()
    kps, scs, des = odisk.extract(own_dense, img, max_num_keypoints=k)
    for i in range(b):
        c = kps[i].shape[0]
        assert c > 0.5 * k or h * w < 8000
        assert torch.equal(pred["keypoints"][i, :c].cpu(), kps[i]) and torch.equal(pred["keypoint_scores"][i, :c].cpu(), scs[i])
        assert (pred["descriptors"][i, :c].cpu() - des[i]).abs().max() < 1e-6
    sd = {n[len("model."):]: v.cpu() for n, v in m.state_dict().items()}
    kps, scs, des = odisk.extract(lambda x: ounet.heatmap_and_dense_descriptors(sd, x), img, max_num_keypoints=k)
    flips = 0
    for i in range(b):
        c = kps[i].shape[0]
        mine = {tuple(p.tolist()): j for j, p in enumerate(pred["keypoints"][i].cpu())}
        for j, p in enumerate(kps[i]):
            jj = mine.get(tuple(p.tolist()))
            if jj is None:
                flips += 1
                continue
            assert abs(pred["keypoint_scores"][i, jj].item() - scs[i][j].item()) < 1e-4 * (1 + abs(scs[i][j].item()))
            assert (pred["descriptors"][i, jj].cpu() - des[i][j]).abs().max() < 1e-4
        assert flips <= max(2, c // 100), (flips, c)
    from parity_utils import record
    record(f"disk_module_native_{b}x{h}x{w}_k{k}", flips=flips)


def test_config5_disk_plus_lightglue_pipeline_vs_oracle():
    """BASELINE config 5 end to end: TwoViewPipeline(extractor = DISK on its native network, matcher = LightGlue with
    128-d input).  The matcher's output against the CPU oracle fed with the extractor's own features (index outputs
    bit-exact, scores within 1e-4), and the extractor's features against the CPU network + stages (checked above)."""
    from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline
    from oracle import lightglue as olg

    # view 1 = view 0 displaced by (16, 16) pixels: the U-Net pools four times (stride 16), so only displacements that are
    # multiples of 16 leave heat-map and descriptors equivariant away from the borders -- the (16, 8) shift of the SuperPoint
    # workloads (stride 8) gave this configuration a weak signal (round 4: 35 matches of 512 key points)
    from glue_factory_colon_amd import synthetic
    h, w, k = 240, 320, 512
    g0, g1 = synthetic.synthetic_pairs(1, h, w, seed=55, dx=16, dy=16)
    img0 = torch.cat([g0 * 0.8, g0, g0 * 0.9], 1).contiguous()
    img1 = torch.cat([g1 * 0.8, g1, g1 * 0.9], 1).contiguous()
    pipe = TwoViewPipeline({
        "extractor": {"name": "extractors.disk_kornia", "weights": "synthetic", "max_num_keypoints": k},
        "matcher": {"name": "matchers.lightglue_pretrained", "features": "disk", "weights": "synthetic",
                    "filter_threshold": 0.1}}).eval().to(DEV)
    size = torch.tensor([[float(w), float(h)]], device=DEV)
    pred = pipe({"view0": {"image": img0.to(DEV), "image_size": size}, "view1": {"image": img1.to(DEV), "image_size": size}})
    assert pred["descriptors0"].shape[-1] == 128 and pred["keypoints0"].shape[1] > k // 2
    sd = {n[len("net."):]: v.cpu() for n, v in pipe.matcher.state_dict().items()}
    assert sd["input_proj.weight"].shape == (256, 128)
    ref = olg.match(sd, pred["keypoints0"].cpu(), pred["keypoints1"].cpu(), pred["descriptors0"].cpu(),
                    pred["descriptors1"].cpu(), size.cpu(), size.cpu(), filter_threshold=0.1)
    assert torch.equal(pred["matches0"].cpu(), ref["matches0"]) and torch.equal(pred["matches1"].cpu(), ref["matches1"])
    assert (pred["matching_scores0"].cpu() - ref["matching_scores0"]).abs().max() < 1e-4
    n_matches = int((ref["matches0"] >= 0).sum())
    # the matches follow the known displacement
    ok = pred["matches0"][0] >= 0
    d = pred["keypoints1"][0][pred["matches0"][0][ok]] - pred["keypoints0"][0][ok]
    on_shift = int(((d - torch.tensor([16.0, 16.0], device=DEV)).abs().max(1).values < 0.5).sum())
    from parity_utils import record
    record("config5_pipeline", matches=n_matches, keypoints=int(pred["keypoints0"].shape[1]), matches_on_the_known_shift=on_shift)
    print(f"config 5 pipeline: {n_matches} matches of {int(pred['keypoints0'].shape[1])} key points, {on_shift} on the known shift")
    assert n_matches >= 60 and on_shift >= 0.6 * n_matches, (n_matches, on_shift)  # measured: 94 matches, 75 on the shift (round 4, (7, 0) shift of white noise: 35)


def test_config5_pair_batched_equals_pair_by_pair():
    """Config 5 in the pair-batched evaluation loop: DISK has no `forward_views` (its own chunked batching), so
    TwoViewPipeline.forward_pairs extracts view by view and still runs the 128-d LightGlue ONCE over all pairs with
    their own key-point counts: integer outputs identical to the pair-by-pair calls, scores within 1e-4."""
    from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline

    pipe = TwoViewPipeline({
        "extractor": {"name": "extractors.disk_kornia", "weights": "synthetic", "max_num_keypoints": 4096},
        "matcher": {"name": "matchers.lightglue_pretrained", "features": "disk", "weights": "synthetic",
                    "filter_threshold": 0.1}}).eval().to(DEV)
    g = torch.Generator().manual_seed(56)
    datas = []
    for h, w in ((160, 208), (176, 240), (160, 208), (208, 160)):
        img0 = torch.rand((1, 3, h, w), generator=g)
        img1 = (img0.roll(5, -1) * 0.9 + 0.05 * torch.rand((1, 3, h, w), generator=g)).contiguous()
        size = torch.tensor([[float(w), float(h)]], device=DEV)
        datas.append({"view0": {"image": img0.to(DEV), "image_size": size}, "view1": {"image": img1.to(DEV), "image_size": size}})
    with torch.no_grad():
        single = [pipe(d) for d in datas]
        multi = pipe.forward_pairs(datas)
    assert len({p["keypoints0"].shape[1] for p in single} | {p["keypoints1"].shape[1] for p in single}) > 1  # ragged
    for a, b in zip(single, multi):
        assert set(a) == set(b)
        for k in ("keypoints0", "keypoints1", "descriptors0", "matches0", "matches1"):
            assert torch.equal(a[k], b[k]), k
        assert (a["matching_scores0"] - b["matching_scores0"]).abs().max() < 1e-4
        assert (a["matches0"] >= 0).sum() > 20


def test_weights_loaded_through_the_wrapper_replace_the_packed_ones():
    """A forward packs device copies of the weights; weights loaded afterwards through a PARENT module's
    load_state_dict (nn.Module recursion, not DiskUnet's own override) must be the ones the next forward uses."""
    m = disk_kornia.DISK({"max_num_keypoints": 128, "weights": "synthetic"}).eval().to(DEV)
    img = torch.rand((1, 3, 64, 96), generator=torch.Generator().manual_seed(3)).to(DEV)
    a = m({"image": img})["keypoint_scores"].clone()
    assert m.model._packed is not None
    other = {"model." + k: v for k, v in weights.disk_state_dict(7).items()}
    missing = m.load_state_dict(other, strict=False)
    assert not missing.unexpected_keys
    assert m.model._packed is None
    b = m({"image": img})["keypoint_scores"]
    assert a.shape != b.shape or not torch.equal(a, b)
    m.load_state_dict({"model." + k: v for k, v in weights.disk_state_dict(0).items()}, strict=False)
    assert torch.equal(m({"image": img})["keypoint_scores"], a)

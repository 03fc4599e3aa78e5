"""The C ABI used WITHOUT Python/torch on the calling side: tests/cabi/cabi_smoke.cpp is compiled with hipcc
against include/gfc_amd.h + libgfc_amd.so, fed a flat weight blob, and must reproduce the key points, scores and
descriptors of the Python boundary module bit for bit (same kernels, same launch sequence); it also checks the
status codes returned for bad arguments."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from glue_factory_colon_amd import superpoint_open, synthetic, weights  # noqa: E402
from glue_factory_colon_amd._superpoint_common import fold_bn  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "glue-factory-colon_amd")


def test_cabi_program_matches_python_module(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "cabi_smoke"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tests", "cabi", "cabi_smoke.cpp"),
                        "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lgfc_amd", f"-Wl,-rpath,{PKG}", "-o", str(exe)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    h, w, k = 96, 136, 200
    img = synthetic.synthetic_images(1, h, w, seed=77)
    sd = weights.superpoint_open_state_dict(0)

    def blk(prefix):
        # folded on the device, as the module does after .cuda() (1/sqrt may round differently on the host)
        a, b = fold_bn(sd[prefix + ".bn.weight"].cuda(), sd[prefix + ".bn.bias"].cuda(),
                       sd[prefix + ".bn.running_mean"].cuda(), sd[prefix + ".bn.running_var"].cuda(), 1e-3)
        return sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"], a.cpu(), b.cpu()

    blob = [struct.pack("ii", h, w), img.numpy().astype(np.float32).tobytes()]
    for bidx in range(4):
        for j in range(2):
            for t in blk(f"backbone.{bidx}.{j}"):
                blob.append(t.float().contiguous().numpy().tobytes())
    det0, des0 = blk("detector.0"), blk("descriptor.0")
    for a, b in zip(det0, des0):
        blob.append(torch.cat([a, b], 0).float().contiguous().numpy().tobytes())
    for t in blk("detector.1"):
        blob.append(t.float().contiguous().numpy().tobytes())
    for t in blk("descriptor.1"):
        blob.append(t.float().contiguous().numpy().tobytes())
    path = tmp_path / "blob.bin"
    path.write_bytes(b"".join(blob))
    r = subprocess.run([str(exe), str(path), str(k)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    lines = r.stdout.strip().splitlines()
    assert "gfx950" in lines[0]
    n = int(lines[1].split()[1])
    rows = np.array([[float(v) for v in ln.split()[1:]] for ln in lines[2:]])
    m = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": k, "detection_threshold": 0.0,
                                    "nms_radius": 3}).eval().cuda()
    p = m({"image": img.cuda()})
    assert n == p["keypoints"].shape[1] == rows.shape[0]
    assert np.array_equal(rows[:, :2], p["keypoints"][0].cpu().numpy().astype(np.float64))
    sc = p["keypoint_scores"][0].cpu().numpy()
    assert np.array_equal(rows[:, 2].astype(np.float32), sc), np.abs(rows[:, 2].astype(np.float32) - sc).max()
    ds = (p["descriptors"][0].cpu().double() * (torch.arange(256) % 7 + 1).double()).sum(-1).numpy()
    assert np.abs(rows[:, 3] - ds).max() < 1e-5


def test_cabi_matcher_program_matches_python_module(tmp_path):
    """The MATCHER through the C ABI alone: tests/cabi/cabi_matcher.cpp (no torch) feeds gfc_lg_forward_ragged with three
    pairs of different key-point counts and the parameter arrays in the layouts `gfc_lg_params` documents; matches0 and
    matching_scores0 must equal LightGlue.forward_pairs bit for bit, and bad arguments must come back as status codes."""
    from glue_factory_colon_amd import lightglue

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "cabi_matcher"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tests", "cabi", "cabi_matcher.cpp"),
                        "-I", os.path.join(ROOT, "include"), "-L", PKG, "-lgfc_amd", f"-Wl,-rpath,{PKG}", "-o", str(exe)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    nl = 3
    model = lightglue.LightGlue({"weights": "synthetic", "filter_threshold": 0.1, "n_layers": nl}).eval().cuda()
    g = torch.Generator().manual_seed(21)
    shapes = [(150, 200), (150, 200), (64, 37)]  # two equal pairs (one batched assignment group) and a ragged one
    items = []
    for m_, n_ in shapes:
        d0 = torch.nn.functional.normalize(torch.randn((1, m_, 256), generator=g), dim=-1)
        d1 = torch.nn.functional.normalize(torch.cat([d0[:, :min(m_, n_)] + 0.2 * torch.randn((1, min(m_, n_), 256), generator=g),
                                                      torch.randn((1, n_ - min(m_, n_), 256), generator=g)], 1), dim=-1)
        size = torch.tensor([[640.0, 480.0]]).cuda()
        items.append({"keypoints0": (torch.rand((1, m_, 2), generator=g) * 400).cuda(),
                      "keypoints1": (torch.rand((1, n_, 2), generator=g) * 400).cuda(),
                      "descriptors0": d0.cuda(), "descriptors1": d1.cuda(),
                      "view0": {"image_size": size}, "view1": {"image_size": size}})
    with torch.no_grad():
        ref = model.forward_pairs(items)
    # the blob: pairs in the order given (equal shapes adjacent, as gfc_lg_forward_ragged's groups want them)
    ms, ns = [a for a, _ in shapes], [b for _, b in shapes]
    kp, de = [], []
    # group 0 = pairs 0, 1 (side-0 rows of both, then side-1 rows of both); group 1 = pair 2
    for grp in ([0, 1], [2]):
        for side in "01":
            for i in grp:
                kp.append(items[i]["keypoints" + side][0].cpu())
                de.append(items[i]["descriptors" + side][0].cpu())
    size = torch.tensor([[640.0, 480.0]] * 3)
    P = model._packed[0]

    by_ptr = {t.data_ptr(): t for t in model._packed[1]}  # the tensors behind the raw pointers of gfc_lg_params

    def arr(ptr, count):  # device array behind a gfc_lg_params pointer -> host floats
        t = by_ptr[int(ptr)]
        assert t.numel() == count, (t.shape, count)
        return t.detach().reshape(-1).cpu()

    blob = [struct.pack("ii", 3, nl), struct.pack("3i", *ms), struct.pack("3i", *ns)]
    f32 = lambda t: t.float().contiguous().numpy().tobytes()  # noqa: E731
    blob += [f32(torch.cat(kp, 0)), f32(torch.cat(de, 0)), f32(size), f32(size), f32(arr(P.posenc_wr, 64))]
    for l in range(nl):
        for name, cnt in (("wqkv", 768 * 256), ("bqkv", 768), ("s_ffn0_w", 512 * 512), ("s_ffn0_b", 512), ("s_ln_g", 512),
                          ("s_ln_b", 512), ("s_ffn3_w", 256 * 512), ("s_ffn3_b", 256), ("c_qkv_w", 512 * 256),
                          ("c_qkv_b", 512), ("c_ffn0_w", 512 * 512), ("c_ffn0_b", 512), ("c_ln_g", 512), ("c_ln_b", 512),
                          ("c_ffn3_w", 256 * 512), ("c_ffn3_b", 256)):
            blob.append(f32(arr(getattr(P, name)[l], cnt)))
        assert not P.s_out_w[l] and not P.c_out_w[l]  # out_proj / to_out are folded into ffn[0]
    for name, cnt in (("final_proj_w", 256 * 256), ("final_proj_b", 256), ("matchability_w", 256), ("matchability_b", 1)):
        blob.append(f32(arr(getattr(P, name)[nl - 1], cnt)))
    path = tmp_path / "matcher.bin"
    path.write_bytes(b"".join(blob))
    r = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-500:])
    lines = r.stdout.strip().splitlines()
    assert "gfx950" in lines[0] and lines[1] == "pairs 3"
    got = {}
    for ln in lines[2:]:
        _, i, j, mm, sc = ln.split()
        got[(int(i), int(j))] = (int(mm), np.float32(float(sc)))
    total = 0
    for i, out in enumerate(ref):
        m0, s0 = out["matches0"][0].cpu().numpy(), out["matching_scores0"][0].cpu().numpy()
        for j in range(len(m0)):
            assert got[(i, j)][0] == m0[j] and got[(i, j)][1] == s0[j], (i, j, got[(i, j)], m0[j], s0[j])
        total += int((m0 >= 0).sum())
    assert total > 5  # (3 layers, random key-point positions: a handful of confident matches)

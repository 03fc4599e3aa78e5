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

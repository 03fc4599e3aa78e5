"""Dump heat-map / descriptor-map checksums and timings of gfc_sp_dense for the loaded library build (A/B of the detector head):
    python tools/micro/dense_dump.py out_a.pt;  GFC_AMD_LIB=tools/ab_libs/libgfc_amd_base.so python tools/micro/dense_dump.py out_b.pt
    python tools/micro/dense_dump.py --compare out_a.pt out_b.pt      -> bit-identical or not"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        if k.startswith("t_"):
            print(f"{k}: {a[k]:.1f} us vs {b[k]:.1f} us")
            continue
        same = torch.equal(a[k], b[k])
        print(f"{k}: {'bit-identical' if same else 'DIFFERENT max |d| = %.3g' % float((a[k] - b[k]).abs().max())}  shape {tuple(a[k].shape)}")
    sys.exit(0)

from glue_factory_colon_amd import _native as nat  # noqa: E402
from glue_factory_colon_amd import superpoint, superpoint_open, synthetic  # noqa: E402

import ctypes  # noqa: E402
if not hasattr(ctypes.CDLL(nat.LIB_PATH), "gfc_sp_detector_head"):  # an older build of the library (A/B): symbol of round 5
    nat.SIGNATURES.pop("gfc_sp_detector_head")

dev = torch.device("cuda", 0)
out = {}
for tag, mod, conf, shape, rgb in (("open_vga", superpoint_open, {}, (8, 480, 640), False),
                                   ("official_ragged", superpoint, {}, (3, 200, 328), True),
                                   ("open_1024", superpoint_open, {}, (2, 1024, 1024), False)):
    m = mod.SuperPoint({"weights": "synthetic", "max_num_keypoints": 512, "detection_threshold": 0.0, "nms_radius": 3, **conf}).eval().to(dev)
    img = synthetic.synthetic_images(shape[0], shape[1], shape[2], seed=11, device=dev)
    if rgb:
        img = torch.cat([img * 0.8, img, img * 0.9], 1).contiguous()
    packed = m.ensure_packed(dev)
    heat, desc = m._runner.dense(packed, img)
    torch.cuda.synchronize()
    out[f"{tag}_heat"], out[f"{tag}_desc"] = heat.cpu(), desc.cpu()
    if tag == "open_vga":
        big = synthetic.synthetic_images(64, 480, 640, seed=5, device=dev)
        for _ in range(2):
            m._runner.dense(packed, big)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m._runner.dense(packed, big)
        e1.record()
        torch.cuda.synchronize()
        out["t_dense_64_vga"] = e0.elapsed_time(e1) * 200
torch.save(out, sys.argv[1])
print("saved", sys.argv[1], {k: (tuple(v.shape) if hasattr(v, "shape") else round(v, 1)) for k, v in out.items()})

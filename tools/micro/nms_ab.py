"""Same-box timing of the NMS (+ candidate emission) of the bench shape: 64 VGA heat-maps, radius 3.
    python tools/micro/nms_ab.py      (GFC_NMS_MODE=1: the LDS-image kernel)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_colon_amd._superpoint_common import SuperPointRunner

dev = torch.device("cuda")
run = SuperPointRunner()
for (b, h, w, k) in ((64, 480, 640, 1024), (64, 1024, 1024, 2048), (2, 480, 640, 1024), (4, 480, 640, 1024), (8, 480, 640, 1024), (12, 480, 640, 1024), (16, 480, 640, 1024), (24, 480, 640, 1024), (32, 480, 640, 1024)):
    heat = torch.rand((b, h, w), device=dev) ** 4
    for name, fn in (("nms (dense map out)", lambda: run.nms(heat, 3, 4)),) + (
            (("nms_select (fused candidates + top-k)", lambda: run.nms_select(heat, 3, 4, None, 0.0, k)),) if b in (2, 64) else ()):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print(f"mode {os.environ.get('GFC_NMS_MODE', '0')}  {b} x {h}x{w}  {name}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us", flush=True)

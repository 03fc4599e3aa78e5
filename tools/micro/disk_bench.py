"""DISK (config 5) on the native network: ms per VGA image of the U-Net and of the whole extractor, and the network's
MFMA rate (algorithmic FLOPs of the nine 5x5 convolutions / time).
    python tools/micro/disk_bench.py [batch]        (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_colon_amd import disk_kornia  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W = 480, 640
dev = torch.device("cuda", 0)
m = disk_kornia.DISK({"weights": "synthetic", "max_num_keypoints": 2048, "force_num_keypoints": True, "chunk": b}).eval().to(dev)
img = torch.rand((b, 3, H, W), device=dev)
layers = [(3, 16, 0), (16, 32, 1), (32, 64, 2), (64, 64, 3), (64, 64, 4), (128, 64, 3), (128, 64, 2), (96, 64, 1), (80, 129, 0)]
flops = sum(2 * 25 * ci * co * (H >> lv) * (W >> lv) for ci, co, lv in layers)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t_net = timed(lambda: m.model.dense_nhwc(img))
t_all = timed(lambda: m({"image": img}))
print(f"DISK U-Net, {b} x VGA: {t_net * 1e3 / b:.3f} ms / image, {flops * b / t_net / 1e12:.1f} TFLOP/s algorithmic "
      f"({flops / 1e9:.1f} GFLOP / image); whole extractor (2048 kpts) {t_all * 1e3 / b:.3f} ms / image")

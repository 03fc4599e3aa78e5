"""Where the from-host leg's time goes: feeder alone (H2D + resize, no model), resize kernel alone, H2D alone."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_colon_amd import synthetic  # noqa: E402
from glue_factory_colon_amd.image_preprocessor import HostImageFeeder, resize  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
raw = synthetic.hpatches_like_host_images(n)
conf = {"resize": 480, "side": "short"}
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    items = list(HostImageFeeder(raw, conf))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"feeder alone: host {1e3 * (t1 - t0) / n:.3f} ms/pair issue, {1e3 * (t2 - t0) / n:.3f} ms/pair to completion")
u8 = raw[0]["view0"]["image"]
d = u8.cuda()
for size in ((480, 640),):
    for _ in range(3):
        resize(d, size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        resize(d, size)
    torch.cuda.synchronize()
    print(f"resize {tuple(u8.shape)} -> {size}: {1e6 * (time.perf_counter() - t0) / 50:.1f} us")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    d = u8.to("cuda", non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print(f"H2D {u8.numel() / 1e6:.2f} MB pinned: {1e6 * dt:.1f} us = {u8.numel() / dt / 1e9:.1f} GB/s")

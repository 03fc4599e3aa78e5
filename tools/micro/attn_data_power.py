"""Is the attention kernel clock / power limited?  Same launch on random and on zero-filled operands (identical
instruction stream, no data-dependent branches except the skipped rescale): the chip's clock is the only difference."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import _native as nat  # noqa: E402

dev = torch.device("cuda", 0)
lib = nat.lib()
st = nat.stream_ptr(dev)
B, K = 32, 1024
R = 2 * B * K
o = torch.empty((R, 256), device=dev)
cross_p = torch.tensor([[i * K, K, (B + i) * K, K] for i in range(B)] + [[(B + i) * K, K, i * K, K] for i in range(B)],
                       dtype=torch.int32, device=dev)
for name, qkv in (("random", torch.randn((R, 768), device=dev)), ("zeros", torch.zeros((R, 768), device=dev)),
                  ("random", torch.randn((R, 768), device=dev)), ("zeros", torch.zeros((R, 768), device=dev))):
    def run():
        nat.check(lib.gfc_attention(nat.ptr(qkv), 768, nat.c_void_p(qkv.data_ptr() + 256 * 4), 768,
                                    nat.c_void_p(qkv.data_ptr() + 512 * 4), 768, nat.ptr(o), 256, nat.ptr(cross_p), 2 * B, K, 4,
                                    0.125, None, 0, st), "attention")
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print(f"{name:7s} operands: {us:7.1f} us per launch, {2 * B * 4 * 2 * 2.0 * K * K * 64 / us / 1e6:6.1f} TFLOP/s")

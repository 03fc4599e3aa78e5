"""Per-wave phase accounting of conv3x3_wino_kernel (diagnostic build -DWINO_DIAG=256, tools/ab_build.sh WORKTREE
wstamps "-DWINO_DIAG=256"): cycles between item start and the end of the K loop, in the column transform + exchange
write + barrier, and in the row transform + stores, summed over a persistent workgroup's items.

    GFC_AMD_LIB=tools/ab_libs/libgfc_amd_wstamps.so python tools/micro/wino_timeline.py
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import _native as nat  # noqa: E402

dev = torch.device("cuda", 0)
lib = nat.lib()
raw = ctypes.CDLL(os.environ["GFC_AMD_LIB"])
raw.gfc_diag_set_wino_stamps.argtypes = [ctypes.c_void_p]
raw.gfc_diag_set_wino_stamps.restype = None
st = nat.stream_ptr(dev)
B = 64
for name, cin, cout, h, w, pool in (("conv2a", 64, 64, 240, 320, 0), ("conv2b", 64, 64, 240, 320, 1),
                                    ("conv3a", 64, 128, 120, 160, 0), ("conv3b", 128, 128, 120, 160, 1),
                                    ("conv4a", 128, 128, 60, 80, 0), ("heads", 128, 512, 60, 80, 0)):
    x = torch.randn((B, h, w, cin), device=dev)
    wt = torch.randn((cout, cin, 3, 3), device=dev) / (3 * cin ** 0.5)
    bias = torch.randn((cout,), device=dev) * 0.1
    ww = torch.empty((16 * cout * cin,), device=dev)
    nat.check(lib.gfc_pack_conv3x3_wino(nat.ptr(wt), nat.ptr(ww), cout, cin, st), "pack")
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    y = torch.empty((B, ho, wo, cout), device=dev)

    def run():
        nat.check(lib.gfc_conv3x3_wino(nat.ptr(x), nat.ptr(ww), nat.ptr(bias), None, None, nat.ptr(y), B, h, w, cin, cout, 1,
                                       pool, st), "wino")

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    nwg = 512
    stamps = torch.zeros((nwg * 4, 8), dtype=torch.int64, device=dev)
    raw.gfc_diag_set_wino_stamps(stamps.data_ptr())
    run()
    torch.cuda.synchronize()
    raw.gfc_diag_set_wino_stamps(None)
    s = stamps.cpu().numpy()
    s = s[s[:, 3] > 0]
    k, xch, sto, n, life = (s[:, i].astype(np.float64) for i in range(5))
    mfma = n * (cin // 16) * 64 * 64
    print(f"{name:7s} {cin:3d}->{cout:3d} @{h}x{w} pool={pool}: {us:8.1f} us; waves {len(s)}, items/wave {np.median(n):.0f}; "
          f"per item: K phase {np.median(k / n):7.0f}  exchange {np.median(xch / n):6.0f}  row+stores {np.median(sto / n):6.0f} "
          f" other {np.median((life - k - xch - sto) / n):6.0f} cycles; MFMA issue {mfma[0] / n[0]:6.0f}/item; "
          f"MFMA/lifetime {np.median(mfma / life):.3f} (x2 waves/SIMD = {2 * np.median(mfma / life):.3f})")

# ---- the stem (conv1a on the VALU + conv1b Winograd + pool), one item per workgroup ----
B, h, w = 64, 480, 640
img = torch.rand((B, h, w), device=dev)
w1 = torch.randn((64, 1, 3, 3), device=dev) / 3
b1, s1, t1 = torch.randn((64,), device=dev) * 0.1, torch.rand((64,), device=dev) + 0.5, torch.randn((64,), device=dev) * 0.1
w2 = torch.randn((64, 64, 3, 3), device=dev) / 24
b2, s2, t2 = torch.randn((64,), device=dev) * 0.1, torch.rand((64,), device=dev) + 0.5, torch.randn((64,), device=dev) * 0.1
w1p = w1.reshape(64, 9).t().contiguous()
w2w = torch.empty((16 * 64 * 64,), device=dev)
nat.check(lib.gfc_pack_conv3x3_wino(nat.ptr(w2), nat.ptr(w2w), 64, 64, st), "pack")
y = torch.empty((B, h // 2, w // 2, 64), device=dev)


def run_stem():
    nat.check(lib.gfc_sp_stem_wino(nat.ptr(img), nat.ptr(w1p), nat.ptr(b1), nat.ptr(s1), nat.ptr(t1), nat.ptr(w2w), nat.ptr(b2),
                                   nat.ptr(s2), nat.ptr(t2), nat.ptr(y), B, h, w, st), "stem")


for _ in range(2):
    run_stem()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    run_stem()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 3
nwg = B * ((h + 15) // 16) * ((w + 7) // 8)
stamps = torch.zeros((nwg * 4, 8), dtype=torch.int64, device=dev)
raw.gfc_diag_set_wino_stamps(stamps.data_ptr())
run_stem()
torch.cuda.synchronize()
raw.gfc_diag_set_wino_stamps(None)
s = stamps.cpu().numpy()
s = s[s[:, 3] > 0]
k, xch, sto, n, life = (s[:, i].astype(np.float64) for i in range(5))
mfma = 4 * 64 * 64.0
print(f"stem    1->64->64 @{h}x{w} pool: {us:8.1f} us; waves {len(s)}; per workgroup: K phase {np.median(k):7.0f}  exchange "
      f"{np.median(xch):6.0f}  row+stores {np.median(sto):6.0f}  prologue + rest {np.median(life - k - xch - sto):6.0f} cycles; "
      f"MFMA issue {mfma:6.0f}; MFMA/lifetime {np.median(mfma / life):.3f} (x2 waves/SIMD = {2 * np.median(mfma / life):.3f})")

"""Per-layer time of conv3x3_wino_kernel at the benchmark's shapes (64 images), for same-box A/B of library builds:
    for l in "" wstag1 wstag2; do GFC_AMD_LIB=${l:+tools/ab_libs/libgfc_amd_$l.so} python tools/micro/wino_layers.py; done
With a -DWINO_DIAG=256 build (GFC_AMD_LIB=...wstamps.so) and `--phase` it also reports, per layer, how far apart the
two persistent workgroups of a CU are when they enter the epilogue of their 10th item (|delta| modulo the item period)."""
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
st = nat.stream_ptr(dev)
phase = "--phase" in sys.argv
raw = None
if phase:
    raw = ctypes.CDLL(os.environ["GFC_AMD_LIB"])
    raw.gfc_diag_set_wino_stamps.argtypes = [ctypes.c_void_p]
    raw.gfc_diag_set_wino_stamps.restype = None
B = 64
total = 0.0
for name, cin, cout, h, w, pool in (("conv2a", 64, 64, 240, 320, 0), ("conv2b", 64, 64, 240, 320, 1),
                                    ("conv3a", 64, 128, 120, 160, 0), ("conv3b", 128, 128, 120, 160, 1),
                                    ("conv4a", 128, 128, 60, 80, 0), ("conv4b", 128, 128, 60, 80, 0),
                                    ("heads", 128, 512, 60, 80, 0)):
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

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 200)
    total += best
    line = f"{name:7s} {cin:3d}->{cout:3d} @{h}x{w} pool={pool}: {best:8.1f} us"
    if phase:
        stamps = torch.zeros((512 * 4, 8), dtype=torch.int64, device=dev)
        raw.gfc_diag_set_wino_stamps(stamps.data_ptr())
        run()
        torch.cuda.synchronize()
        raw.gfc_diag_set_wino_stamps(None)
        s = stamps.cpu().numpy().reshape(512, 4, 8)[:, 0]  # wave 0 of every workgroup
        ok = s[:, 3] >= 10
        s = s[ok]
        period = np.median((s[:, 0] + s[:, 1] + s[:, 2]) / s[:, 3])
        cu = (s[:, 5] >> 8) & 0xF | (((s[:, 5] >> 12) & 0x1) << 4) | (((s[:, 5] >> 13) & 0x7) << 5) | ((s[:, 5] >> 32) << 8)
        deltas = []
        for key in np.unique(cu):
            t = s[cu == key, 7]
            if len(t) == 2:
                d = abs(int(t[0]) - int(t[1])) % period
                deltas.append(min(d, period - d) / period)
        deltas = np.array(deltas)
        line += (f"; item period {period:.0f} cycles; CUs with two workgroups {len(deltas)}; epilogue phase offset / period: "
                 f"median {np.median(deltas):.3f}, p10 {np.percentile(deltas, 10):.3f}, p90 {np.percentile(deltas, 90):.3f} "
                 f"(0 = epilogues coincide, 0.5 = perfectly interleaved)")
    print(line, flush=True)
print(f"sum over the seven launches: {total:.1f} us  (lib = {os.environ.get('GFC_AMD_LIB', 'worktree')})")

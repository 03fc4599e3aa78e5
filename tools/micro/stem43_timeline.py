"""Per-wave phase accounting of stem_wino43_kernel (diagnostic build: tools/ab_build.sh WORKTREE s4 "-DS4_DIAG=1").
    GFC_AMD_LIB=tools/ab_libs/libgfc_amd_s4.so python tools/micro/stem43_timeline.py
Cycles per item and wave: image patch -> LDS + barrier | conv1a tile 0 | barrier | conv1a tile 1 (inside chunk 1) |
the four chunks' k groups | chunk barriers | epilogue (column pass, two exchange passes)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_colon_amd import _native as nat

dev = torch.device("cuda", 0)
lib = nat.lib()
raw = ctypes.CDLL(os.environ["GFC_AMD_LIB"])
raw.gfc_diag_set_stem43_stamps.argtypes = [ctypes.c_void_p]
raw.gfc_diag_set_stem43_stamps.restype = None
st = nat.stream_ptr(dev)
B, H, W = 64, 480, 640
img = torch.rand((B, H, W), device=dev)
w1 = torch.randn((9, 64), device=dev) / 3
b1, s1, t1 = torch.randn((64,), device=dev) * 0.1, torch.rand((64,), device=dev) + 0.5, torch.randn((64,), device=dev) * 0.1
w2 = torch.randn((64, 64, 3, 3), device=dev) / 24
w43 = torch.empty((36 * 64 * 64,), device=dev)
nat.check(lib.gfc_pack_conv3x3_wino43(nat.ptr(w2), nat.ptr(w43), 64, 64, st), "pack")
y = torch.empty((B, H // 2, W // 2, 64), device=dev)


def run():
    nat.check(lib.gfc_sp_stem_wino43(nat.ptr(img), nat.ptr(w1), nat.ptr(b1), nat.ptr(s1), nat.ptr(t1), nat.ptr(w43), nat.ptr(b1),
                                     nat.ptr(s1), nat.ptr(t1), nat.ptr(y), B, H, W, st), "stem43")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 200
stamps = torch.zeros((256 * 12, 16), dtype=torch.int64, device=dev)
raw.gfc_diag_set_stem43_stamps(stamps.data_ptr())
run()
torch.cuda.synchronize()
raw.gfc_diag_set_stem43_stamps(None)
s = stamps.cpu().numpy().astype(np.float64)
s = s[s[:, 7] > 0]
n = s[:, 7]
names = ["item head + seams", "half: transform", "half: MFMA issue", "conv1a units", "k groups", "chunk barriers", "epilogue"]
print(f"stem43: {us:.1f} us per launch; waves {len(s)}, items/wave {np.median(n):.0f}; lifetime/item {np.median(s[:, 8] / n):.0f} cycles; "
      f"MFMA issue per wave and item: {(192 + 10) * 64} cycles (x3 waves per SIMD = {(192 + 10) * 64 * 3})")
for i, nm in enumerate(names):
    print(f"  {nm:15s} {np.median(s[:, i] / n):8.0f}   (p10 {np.percentile(s[:, i] / n, 10):8.0f}  p90 {np.percentile(s[:, i] / n, 90):8.0f})")
print(f"  conv1a units: of which LDS reads + the five MFMAs {np.median(s[:, 9] / n):8.0f} (p90 {np.percentile(s[:, 9] / n, 90):8.0f})")
for xi in range(6):
    sel = s[(np.arange(len(s)) % 12) % 6 == xi]
    print(f"  xi={xi}: k groups {np.median(sel[:, 4] / sel[:, 7]):8.0f}  (16 halves: transform {np.median(sel[:, 1] / sel[:, 7]):8.0f}, MFMA issue "
          f"{np.median(sel[:, 2] / sel[:, 7]):8.0f})  chunk barriers {np.median(sel[:, 5] / sel[:, 7]):8.0f}")

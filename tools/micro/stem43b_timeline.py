"""Per-wave phase accounting of stem_wino43b_kernel (diagnostic build: tools/ab_build.sh WORKTREE b4t "-DB4_DIAG=32").
    GFC_AMD_LIB=tools/ab_libs/libgfc_amd_b4t.so python tools/micro/stem43b_timeline.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_colon_amd import _native as nat

dev = torch.device("cuda", 0)
lib = nat.lib()
raw = ctypes.CDLL(os.environ["GFC_AMD_LIB"])
raw.gfc_diag_set_stem43b_stamps.argtypes = [ctypes.c_void_p]
B, H, W = 64, 480, 640
g = torch.Generator().manual_seed(1)
img = torch.rand((B, H, W), generator=g).to(dev)
w1 = (torch.randn((9, 64), generator=g) / 3).to(dev)
b1 = (torch.randn((64,), generator=g) * 0.1).to(dev)
w2 = (torch.randn((64, 64, 3, 3), generator=g) / 24).to(dev)
b2 = (torch.randn((64,), generator=g) * 0.1).to(dev)
s = (torch.rand((64,), generator=g) + 0.5).to(dev)
t = (torch.randn((64,), generator=g) * 0.1).to(dev)
st = nat.stream_ptr(dev)
wp = torch.empty((36 * 64 * 64,), device=dev)
nat.check(lib.gfc_pack_conv3x3_wino43b(nat.ptr(w2), nat.ptr(wp), 64, 64, st), "p")
y = torch.empty((B, H // 2, W // 2, 64), device=dev)
run = lambda: nat.check(lib.gfc_sp_stem_wino43b(nat.ptr(img), nat.ptr(w1), nat.ptr(b1), nat.ptr(s), nat.ptr(t), nat.ptr(wp), nat.ptr(b2),
                                                nat.ptr(s), nat.ptr(t), nat.ptr(y), B, H, W, st), "stem")
for _ in range(3):
    run()
stamps = torch.zeros((256 * 8, 8), dtype=torch.int64, device=dev)
raw.gfc_diag_set_stem43b_stamps(stamps.data_ptr())
run()
torch.cuda.synchronize()
raw.gfc_diag_set_stem43b_stamps(None)
d = stamps.cpu().numpy().astype(np.float64)
d = d[d[:, 6] > 0]
n = d[:, 6]
names = ["item prologue (image, conv1a 0)", "P2 transform + filter wait + barrier", "P3a (36 MFMAs + conv1a)", "P3b (36 MFMAs)",
         "barriers closing P3a/P3b", "epilogue"]
per_item = [d[:, 0] / n, d[:, 1] / n, d[:, 2] / n, d[:, 3] / n, d[:, 4] / n, d[:, 5] / n]
tot = d[:, 7] / n
for nm, v in zip(names, per_item):
    lo = np.median(v[d.shape[0] // 2:]) if False else np.median(v)
    print(f"{nm:40s} {np.median(v):9.0f} cycles / item   (waves 0-3: {np.median(v.reshape(-1, 8)[:, :4]):9.0f}, waves 4-7: {np.median(v.reshape(-1, 8)[:, 4:]):9.0f})")
print(f"{'whole item':40s} {np.median(tot):9.0f} cycles;  MFMA pipe time per SIMD and item: {2 * 8 * 72 * 32} cycles")

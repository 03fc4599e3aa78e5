"""Same-box timing of the two stem kernels (F(2x2,3x3) vs F(4x4,3x3)) on the bench shape: 64 VGA images.
    python tools/micro/stem_ab.py [B H W]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_colon_amd import _native as nat

B, H, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 480, 640)
lib = nat.lib()
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
img = torch.rand((B, H, W), generator=g).to(dev)
w1 = (torch.randn((9, 64), generator=g) / 3).to(dev)
b1 = (torch.randn((64,), generator=g) * 0.1).to(dev)
w2 = (torch.randn((64, 64, 3, 3), generator=g) / 24).to(dev)
b2 = (torch.randn((64,), generator=g) * 0.1).to(dev)
s = (torch.rand((64,), generator=g) + 0.5).to(dev)
t = (torch.randn((64,), generator=g) * 0.1).to(dev)
st = nat.stream_ptr(dev)
w23 = torch.empty((16 * 64 * 64,), device=dev)
w43 = torch.empty((36 * 64 * 64,), device=dev)
nat.check(lib.gfc_pack_conv3x3_wino(nat.ptr(w2), nat.ptr(w23), 64, 64, st), "p")
nat.check(lib.gfc_pack_conv3x3_wino43(nat.ptr(w2), nat.ptr(w43), 64, 64, st), "p")
y = {k: torch.empty((B, H // 2, W // 2, 64), device=dev) for k in ("f23", "f43")}
fn = {"f23": (lib.gfc_sp_stem_wino, w23), "f43": (lib.gfc_sp_stem_wino43, w43)}
for rep in range(3):
    for k, (f, wp) in fn.items():
        for _ in range(3):
            nat.check(f(nat.ptr(img), nat.ptr(w1), nat.ptr(b1), nat.ptr(s), nat.ptr(t), nat.ptr(wp), nat.ptr(b2), nat.ptr(s),
                        nat.ptr(t), nat.ptr(y[k]), B, H, W, st), k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            nat.check(f(nat.ptr(img), nat.ptr(w1), nat.ptr(b1), nat.ptr(s), nat.ptr(t), nat.ptr(wp), nat.ptr(b2), nat.ptr(s),
                        nat.ptr(t), nat.ptr(y[k]), B, H, W, st), k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{k}: {ms:.3f} ms per launch of {B} images {H}x{W}", flush=True)
print("max |f43 - f23| =", float((y["f43"] - y["f23"]).abs().max()))

"""Per-wave cycle stamps of gemm_nt_kernel (diagnostic build -DGEMM_DIAG=16, tools/ab_build.sh WORKTREE stamps "-DGEMM_DIAG=16"):
where do the cycles of a K = 256 GEMM go -- prologue, K loop, epilogue, store drain -- and how busy is each SIMD?

    GFC_AMD_LIB=tools/ab_libs/libgfc_amd_stamps.so python tools/micro/gemm_timeline.py [N] [K] [M]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import _native as nat  # noqa: E402

ROT = len(sys.argv) > 1 and sys.argv[1] == "rot"  # QKV GEMM: N = 768, rotary epilogue on the first 512 columns
if ROT:
    sys.argv = [sys.argv[0], "768", "256"] + sys.argv[2:]
FFN = len(sys.argv) > 1 and sys.argv[1] == "ffn"  # the row-owning ffn[0] + LayerNorm + GELU kernel (N = 512, K = 256 + 256)
if FFN:
    sys.argv = [sys.argv[0], "512", "512"] + sys.argv[2:]
MLP = len(sys.argv) > 1 and sys.argv[1] == "mlp"  # the whole-FFN kernel gfc_ffn_fused (round 4): phases of one 128-row item
if MLP:
    FFN = True
    sys.argv = [sys.argv[0], "512", "512"] + sys.argv[2:]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = torch.device("cuda", 0)
lib = nat.lib()
raw = ctypes.CDLL(os.environ["GFC_AMD_LIB"])
raw.gfc_diag_set_gemm_stamps.argtypes = [ctypes.c_void_p]
raw.gfc_diag_set_gemm_stamps.restype = None
st = nat.stream_ptr(dev)
A = torch.randn((M, K), device=dev)
W = torch.randn((N, K), device=dev) / K ** 0.5
b = torch.randn((N,), device=dev)
Y = torch.empty((M, N), device=dev)


gamma, beta = torch.rand((N,), device=dev) + 0.5, torch.randn((N,), device=dev)
A1 = torch.randn((M, K // 2), device=dev)


cos_t, sin_t = torch.rand((M, 64), device=dev), torch.rand((M, 64), device=dev)
W3, b3 = torch.randn((256, 512), device=dev) / 512 ** 0.5, torch.randn((256,), device=dev)
Y2 = torch.empty((M, 256), device=dev)


def run():
    if ROT:
        nat.check(lib.gfc_linear(nat.ptr(A), K, K, None, 0, 0, nat.ptr(W), K, nat.ptr(b), None, None, 1.0, None, nat.ptr(cos_t),
                                 nat.ptr(sin_t), 512, nat.ptr(Y), N, M, N, st), "linear")
        return
    if MLP:
        nat.check(lib.gfc_ffn_fused(nat.ptr(A), K, K // 2, nat.ptr(A1), K // 2, K // 2, nat.ptr(W), K, nat.ptr(b),
                                    nat.ptr(gamma), nat.ptr(beta), nat.ptr(W3), 512, nat.ptr(b3), nat.ptr(A1), nat.ptr(Y2), 256,
                                    M, st), "mlp")
        return
    if FFN:
        nat.check(lib.gfc_linear_layernorm_gelu(nat.ptr(A), K, K // 2, nat.ptr(A1), K // 2, K // 2, nat.ptr(W), K, nat.ptr(b),
                                                nat.ptr(gamma), nat.ptr(beta), nat.ptr(Y), N, M, N, st), "ffn")
        return
    nat.check(lib.gfc_linear(nat.ptr(A), K, K, None, 0, 0, nat.ptr(W), K, nat.ptr(b), None, None, 1.0, None, None, None, 0,
                             nat.ptr(Y), N, M, N, st), "linear")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"M={M} N={N} K={K}: {us:.1f} us  {2 * M * N * K / us / 1e6:.1f} TFLOP/s")
nwg = ((M + 127) // 128) * (1 if FFN else (N + 127) // 128)
WPW = 8 if FFN else 4  # waves per workgroup
stamps = torch.zeros((nwg * WPW, 8), dtype=torch.int64, device=dev)
raw.gfc_diag_set_gemm_stamps(stamps.data_ptr())
run()
torch.cuda.synchronize()
raw.gfc_diag_set_gemm_stamps(None)
s = stamps.cpu().numpy().astype(np.int64)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", f"gemm_stamps_N{N}_K{K}.npy"), s)
t = s[:, :5]
hw, xcc = s[:, 5], s[:, 6] & 0xF
if MLP:  # stamps: 0 entry, 1 / 2 first K loop, 3 LayerNorm + GELU done, 4 second K loop done, 7 stores issued
    t6 = np.concatenate([s[:, :5], s[:, 7:8]], axis=1)
    d6 = np.diff(t6, axis=1)
    for name, col in zip(("prologue", "k loop 1", "stats+LN+GELU", "dump + k loop 2", "residual + stores"), range(5)):
        v = d6[:, col]
        print(f"{name:18s} median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
    life6 = t6[:, 5] - t6[:, 0]
    print(f"wave lifetime median {np.median(life6):.0f}; MFMA cycles per wave {(2048 + 1024) * 64} (first GEMM 131072, second 65536)")
    sys.exit(0)
print("waves", len(s), "stamped", int((t[:, 0] > 0).sum()))
d = np.diff(t, axis=1)
for name, col in zip(("prologue", "k loop", "statistics" if FFN else "epilogue", "gelu + stores" if FFN else "store drain"), range(4)):
    v = d[:, col]
    print(f"{name:12s} median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
life = t[:, 4] - t[:, 0]
MF = K // 2 * (8 if FFN else 4) * 64  # MFMA issue cycles per wave
print(f"wave lifetime median {np.median(life):.0f}; MFMA cycles per wave {MF}")
simd = (hw >> 4) & 3
cu = (xcc << 16) | ((hw >> 8) & 0xFF)
keys = cu * 4 + simd
util, cover, spans, nw = [], [], [], []
for k in np.unique(keys):
    m = keys == k
    tt = t[m]
    span = tt[:, 4].max() - tt[:, 0].min()
    iv = sorted((a, b2) for a, b2 in zip(tt[:, 1], tt[:, 2]))
    covered, cur_a, cur_b = 0, None, None
    for a, b2 in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                covered += cur_b - cur_a
            cur_a, cur_b = a, b2
        else:
            cur_b = max(cur_b, b2)
    covered += cur_b - cur_a
    mf = m.sum() * MF
    util.append(mf / span)
    cover.append(covered / span)
    spans.append(span)
    nw.append(m.sum())
print(f"SIMDs {len(util)}  waves/SIMD median {np.median(nw):.0f}  span median {np.median(spans):.0f} cycles")
print(f"MFMA cycles / span: median {np.median(util):.3f}  (p10 {np.percentile(util, 10):.3f}, p90 {np.percentile(util, 90):.3f})")
print(f"time with >= 1 wave inside its K loop / span: median {np.median(cover):.3f}")
print(f"MFMA cycles / time covered by K loops: median {np.median(np.array(util) / np.array(cover)):.3f}")
# one SIMD's timeline
k = np.unique(keys)[len(np.unique(keys)) // 2]
m = keys == k
tt = t[m]
o = np.argsort(tt[:, 0])
base = tt[:, 0].min()
print("one SIMD (entry, k loop start, k loop end, epilogue issued, drained) relative cycles:")
for r in tt[o]:
    print("  ", " ".join(f"{int(x - base):8d}" for x in r))

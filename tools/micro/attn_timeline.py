"""Per-wave cycle accounting of attention_kernel<2,4> (diagnostic build -DATT_DIAG, tools/ab_build.sh WORKTREE astamps
"-DATT_DIAG"): prologue (Q fragments, first K/V tile), key loop, of which: waiting for
the staged next tile (global loads + LDS writes) and at the workgroup barrier, epilogue.

    GFC_AMD_LIB=tools/ab_libs/libgfc_amd_astamps.so python tools/micro/attn_timeline.py
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
raw.gfc_diag_set_attn_stamps.argtypes = [ctypes.c_void_p]
raw.gfc_diag_set_attn_stamps.restype = None
st = nat.stream_ptr(dev)
B, K = 32, 1024
R = 2 * B * K
qkv = torch.randn((R, 768), device=dev)
o = torch.empty((R, 256), device=dev)
cross_p = torch.tensor([[i * K, K, (B + i) * K, K] for i in range(B)] + [[(B + i) * K, K, i * K, K] for i in range(B)],
                       dtype=torch.int32, device=dev)


def run():
    nat.check(lib.gfc_attention(nat.ptr(qkv), 768, nat.c_void_p(qkv.data_ptr() + 256 * 4), 768,
                                nat.c_void_p(qkv.data_ptr() + 512 * 4), 768, nat.ptr(o), 256, nat.ptr(cross_p), 2 * B, K, 4,
                                0.125, None, 0, st), "attention")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print(f"cross attention: {e0.elapsed_time(e1) * 100:.1f} us per launch (diagnostic build)")
nwg = 4 * 4 * 2 * B
stamps = torch.zeros((nwg * 4, 8), dtype=torch.int64, device=dev)
raw.gfc_diag_set_attn_stamps(stamps.data_ptr())
run()
torch.cuda.synchronize()
raw.gfc_diag_set_attn_stamps(None)
s = stamps.cpu().numpy().astype(np.float64)
s = s[s[:, 1] > 0]
pro, loop, bar, sto, epi = (s[:, i] for i in range(5))
mfma = 16 * 256 * 64.0
print(f"waves {len(s)}; per wave (median cycles): prologue {np.median(pro):.0f}, key loop {np.median(loop):.0f} "
      f"(staging wait + LDS write {np.median(sto):.0f}, barrier {np.median(bar):.0f}), epilogue {np.median(epi):.0f}; "
      f"MFMA issue {mfma:.0f}")
life = pro + loop + epi
print(f"2 x MFMA / lifetime = {2 * mfma / np.median(life):.3f}; 2 x MFMA / (loop - barrier - staging) = "
      f"{2 * mfma / np.median(loop - bar - sto):.3f}")
hw = s[:, 5].astype(np.int64)
t0 = s[:, 6]
cu = (hw >> 8) & 0xFF
print("entry-time spread of the first round (cycles, p10/p50/p90 of t_entry - min):",
      np.percentile(t0 - t0.min(), [10, 50, 90]).round(0))

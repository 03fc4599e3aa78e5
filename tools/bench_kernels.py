"""Micro-benchmarks of the individual kernels at the shapes of the benchmark configuration
(32 VGA pairs, 1024 keypoints), through the C ABI.  Run on the GPU box:
    python tools/bench_kernels.py [--only gemm|conv|attn]
Prints one line per case: time, TFLOP/s, fraction of the fp32-MFMA peak."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import _native as nat  # noqa: E402

DEV = torch.device("cuda", 0)
PEAK = 157.3


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, secs, flops):
    tf = flops / secs / 1e12
    print(f"{name:58s} {secs * 1e6:9.1f} us  {tf:7.1f} TFLOP/s  {tf / PEAK * 100:5.1f} %", flush=True)


def bench_gemm():
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    R = 65536
    x = torch.randn((R, 256), device=DEV)
    msg = torch.randn((R, 256), device=DEV)
    h = torch.randn((R, 512), device=DEV)
    cos = torch.rand((R, 64), device=DEV)
    sin = torch.rand((R, 64), device=DEV)
    cases = [("qkv  N=768 K=256 rotary", x, None, 768, 256, 0, True, False),
             ("qkv' N=512 K=256", x, None, 512, 256, 0, False, False),
             ("out  N=256 K=256", x, None, 256, 256, 0, False, False),
             ("ffn0 N=512 K=256+256 (concat)", x, msg, 512, 256, 256, False, False),
             ("ffn3 N=256 K=512 residual", h, None, 256, 512, 0, False, True),
             ("qkv  N=768 K=256 (no rotary)", x, None, 768, 256, 0, False, False)]
    for name, a0, a1, n, k0, k1, rot, res in cases:
        w = torch.randn((n, k0 + k1), device=DEV) / 16
        b = torch.randn((n,), device=DEV)
        y = torch.empty((R, n), device=DEV)
        resid = torch.randn((R, n), device=DEV) if res else None

        def fn():
            nat.check(lib.gfc_linear(nat.ptr(a0), a0.shape[1], k0, nat.ptr(a1), 0 if a1 is None else a1.shape[1], k1,
                                     nat.ptr(w), k0 + k1, nat.ptr(b), None, None, 1.0, nat.ptr(resid),
                                     nat.ptr(cos) if rot else None, nat.ptr(sin) if rot else None, 512 if rot else 0,
                                     nat.ptr(y), n, R, n, st), "linear")

        report(f"gemm[{os.environ.get('GFC_GEMM_TILE', 'auto')}] {name}", timeit(fn), 2.0 * R * n * (k0 + k1))


def bench_ffn_fused():
    """ffn[0] -> LayerNorm -> GELU: GEMM + in-place LayerNorm pass vs the row-owning fused kernel."""
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    R = 65536
    x, msg = torch.randn((R, 256), device=DEV), torch.randn((R, 256), device=DEV)
    w = torch.randn((512, 512), device=DEV) / 22
    b, ga, be = torch.randn((512,), device=DEV), torch.rand((512,), device=DEV) + 0.5, torch.randn((512,), device=DEV)
    y = torch.empty((R, 512), device=DEV)

    def two():
        nat.check(lib.gfc_linear(nat.ptr(x), 256, 256, nat.ptr(msg), 256, 256, nat.ptr(w), 512, nat.ptr(b), None, None, 1.0,
                                 None, None, None, 0, nat.ptr(y), 512, R, 512, st), "linear")
        nat.check(lib.gfc_layernorm_gelu(nat.ptr(y), 512, R, 512, nat.ptr(ga), nat.ptr(be), st), "ln")

    def one():
        nat.check(lib.gfc_linear_layernorm_gelu(nat.ptr(x), 256, 256, nat.ptr(msg), 256, 256, nat.ptr(w), 512, nat.ptr(b),
                                                nat.ptr(ga), nat.ptr(be), nat.ptr(y), 512, R, 512, st), "fused")

    report("ffn0 + layernorm_gelu (2 kernels)", timeit(two), 2.0 * R * 512 * 512)
    report("ffn0+LN+GELU fused (rows512 kernel)", timeit(one), 2.0 * R * 512 * 512)
    # the whole FFN: ffn0 -> LN -> GELU -> ffn3 + residual, two kernels vs gfc_ffn_fused (round 4)
    w3 = torch.randn((256, 512), device=DEV) / 22
    b3 = torch.randn((256,), device=DEV)
    out = torch.empty((R, 256), device=DEV)

    def ffn3():
        nat.check(lib.gfc_linear(nat.ptr(y), 512, 512, None, 0, 0, nat.ptr(w3), 512, nat.ptr(b3), None, None, 1.0,
                                 nat.ptr(x), None, None, 0, nat.ptr(out), 256, R, 256, st), "ffn3")

    def pair():
        one()
        ffn3()

    def mlp():
        nat.check(lib.gfc_ffn_fused(nat.ptr(x), 256, 256, nat.ptr(msg), 256, 256, nat.ptr(w), 512, nat.ptr(b), nat.ptr(ga),
                                    nat.ptr(be), nat.ptr(w3), 512, nat.ptr(b3), nat.ptr(x), nat.ptr(out), 256, R, st), "mlp")

    flops = 2.0 * R * 512 * 512 + 2.0 * R * 512 * 256
    for _ in range(2):  # alternated: same box, same thermal state
        report("ffn3 N=256 K=512 residual (gemm_nt)", timeit(ffn3), 2.0 * R * 512 * 256)
        report("FFN as two kernels (rows512 + ffn3)", timeit(pair), flops)
        report("FFN whole MLP in one kernel (gfc_ffn_fused)", timeit(mlp), flops)


def bench_gemm_sweep():
    """Fixed vs per-K cost: N = 256, K = 256..2048 (synthetic shapes)."""
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    R = 65536
    for n in (256, 512):
        for k in (256, 512, 1024, 2048):
            a = torch.randn((R, k), device=DEV)
            w = torch.randn((n, k), device=DEV) / 16
            b = torch.randn((n,), device=DEV)
            y = torch.empty((R, n), device=DEV)

            def fn():
                nat.check(lib.gfc_linear(nat.ptr(a), k, k, None, 0, 0, nat.ptr(w), k, nat.ptr(b), None, None, 1.0,
                                         None, None, None, 0, nat.ptr(y), n, R, n, st), "linear")

            report(f"gemm[{os.environ.get('GFC_GEMM_TILE', 'auto')}] sweep N={n} K={k}", timeit(fn), 2.0 * R * n * k)


def bench_gemm_msweep():
    """Fixed launch cost vs per-tile cost: N = 256, K = 256, M = 8k .. 512k rows."""
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    n = k = 256
    for m in (8192, 16384, 32768, 65536, 131072, 262144, 524288):
        a = torch.randn((m, k), device=DEV)
        w = torch.randn((n, k), device=DEV) / 16
        b = torch.randn((n,), device=DEV)
        y = torch.empty((m, n), device=DEV)

        def fn():
            nat.check(lib.gfc_linear(nat.ptr(a), k, k, None, 0, 0, nat.ptr(w), k, nat.ptr(b), None, None, 1.0, None,
                                     None, None, 0, nat.ptr(y), n, m, n, st), "linear")

        report(f"gemm msweep M={m} N=256 K=256", timeit(fn), 2.0 * m * n * k)


def bench_gemm_small():
    """Batch-1 shapes (M = 2048 rows): GFC_GEMM_TILE 3 = 64x64 tiles, 4 = 128x128 tiles, unset = by problem size."""
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    for m in (2048, 8192):
        for n, k in ((768, 256), (512, 512), (256, 512)):
            a = torch.randn((m, k), device=DEV)
            w = torch.randn((n, k), device=DEV) / 16
            b = torch.randn((n,), device=DEV)
            y = torch.empty((m, n), device=DEV)

            def fn():
                nat.check(lib.gfc_linear(nat.ptr(a), k, k, None, 0, 0, nat.ptr(w), k, nat.ptr(b), None, None, 1.0, None,
                                         None, None, 0, nat.ptr(y), n, m, n, st), "linear")

            report(f"gemm[tile {os.environ.get('GFC_GEMM_TILE', 'auto')}] M={m} N={n} K={k}", timeit(fn, iters=50),
                   2.0 * m * n * k)


def bench_conv():
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    B = 32
    for name, h, w, cin, cout, pool in [("conv1b 64->64 @480x640 +pool", 480, 640, 64, 64, 1),
                                        ("conv2a 64->64 @240x320", 240, 320, 64, 64, 0),
                                        ("conv2b 64->64 @240x320 +pool", 240, 320, 64, 64, 1),
                                        ("conv3a 64->128 @120x160", 120, 160, 64, 128, 0),
                                        ("conv3b 128->128 @120x160 +pool", 120, 160, 128, 128, 1),
                                        ("conv4a 128->128 @60x80", 60, 80, 128, 128, 0),
                                        ("heads 128->512 @60x80", 60, 80, 128, 512, 0)]:
        x = torch.randn((B, h, w, cin), device=DEV)
        wt = torch.randn((9, cout, cin), device=DEV) / (3 * cin ** 0.5)
        bias = torch.randn((cout,), device=DEV)
        sc = torch.rand((cout,), device=DEV) + 0.5
        sh = torch.randn((cout,), device=DEV)
        y = torch.empty((B, h // 2 if pool else h, w // 2 if pool else w, cout), device=DEV)

        def fn():
            nat.check(lib.gfc_conv3x3(nat.ptr(x), nat.ptr(wt), nat.ptr(bias), nat.ptr(sc), nat.ptr(sh), nat.ptr(y), B,
                                      h, w, cin, cout, 1, pool, st), "conv")

        report(name, timeit(fn, iters=10), 2.0 * 9 * B * h * w * cin * cout)


def bench_conv_wino():
    """Winograd F(2x2,3x3) fp32-MFMA convolution at the shapes of bench_conv; TFLOP/s are ALGORITHMIC (direct-
    convolution FLOPs / time), the matrix pipe executes 4/9 of them."""
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    B = 32
    for name, h, w, cin, cout, pool in [("wino conv1b 64->64 @480x640 +pool", 480, 640, 64, 64, 1),
                                        ("wino conv2a 64->64 @240x320", 240, 320, 64, 64, 0),
                                        ("wino conv2b 64->64 @240x320 +pool", 240, 320, 64, 64, 1),
                                        ("wino conv3a 64->128 @120x160", 120, 160, 64, 128, 0),
                                        ("wino conv3b 128->128 @120x160 +pool", 120, 160, 128, 128, 1),
                                        ("wino conv4a 128->128 @60x80", 60, 80, 128, 128, 0),
                                        ("wino heads 128->512 @60x80", 60, 80, 128, 512, 0)]:
        x = torch.randn((B, h, w, cin), device=DEV)
        wt = torch.randn((cout, cin, 3, 3), device=DEV) / (3 * cin ** 0.5)
        ww = torch.empty((16 * cout * cin,), device=DEV)
        nat.check(lib.gfc_pack_conv3x3_wino(nat.ptr(wt), nat.ptr(ww), cout, cin, st), "pack")
        bias = torch.randn((cout,), device=DEV)
        sc = torch.rand((cout,), device=DEV) + 0.5
        sh = torch.randn((cout,), device=DEV)
        y = torch.empty((B, h // 2 if pool else h, w // 2 if pool else w, cout), device=DEV)

        def fn():
            nat.check(lib.gfc_conv3x3_wino(nat.ptr(x), nat.ptr(ww), nat.ptr(bias), nat.ptr(sc), nat.ptr(sh), nat.ptr(y),
                                           B, h, w, cin, cout, 1, pool, st), "conv_wino")

        report(name, timeit(fn, iters=10), 2.0 * 9 * B * h * w * cin * cout)
    B, h, w = 64, 480, 640
    img = torch.rand((B, h, w), device=DEV)
    w1 = torch.randn((9, 64), device=DEV) / 3
    w2 = torch.randn((64, 64, 3, 3), device=DEV) / 24
    w2w = torch.empty((16 * 64 * 64,), device=DEV)
    nat.check(lib.gfc_pack_conv3x3_wino(nat.ptr(w2), nat.ptr(w2w), 64, 64, st), "pack")
    b1, b2 = torch.randn((64,), device=DEV), torch.randn((64,), device=DEV)
    s1, s2 = torch.rand((64,), device=DEV) + 0.5, torch.rand((64,), device=DEV) + 0.5
    t1, t2 = torch.randn((64,), device=DEV), torch.randn((64,), device=DEV)
    y = torch.empty((B, h // 2, w // 2, 64), device=DEV)

    def fn2():
        nat.check(lib.gfc_sp_stem_wino(nat.ptr(img), nat.ptr(w1), nat.ptr(b1), nat.ptr(s1), nat.ptr(t1), nat.ptr(w2w),
                                       nat.ptr(b2), nat.ptr(s2), nat.ptr(t2), nat.ptr(y), B, h, w, st), "stem_wino")

    report("wino stem conv1a+conv1b+pool @480x640 x64", timeit(fn2, iters=10), 2.0 * 9 * B * h * w * (64 + 64 * 64))


def bench_stem():
    """The dominant kernel: conv1a + conv1b + pool in one launch (gfc_sp_stem), 64 VGA images as in bench.py."""
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    B, h, w = 64, 480, 640
    img = torch.rand((B, h, w), device=DEV)
    w1 = torch.randn((9, 64), device=DEV) / 3
    w2 = torch.randn((9, 64, 64), device=DEV) / 24
    b1, b2 = torch.randn((64,), device=DEV), torch.randn((64,), device=DEV)
    s1, s2 = torch.rand((64,), device=DEV) + 0.5, torch.rand((64,), device=DEV) + 0.5
    t1, t2 = torch.randn((64,), device=DEV), torch.randn((64,), device=DEV)
    y = torch.empty((B, h // 2, w // 2, 64), device=DEV)

    def fn():
        nat.check(lib.gfc_sp_stem(nat.ptr(img), nat.ptr(w1), nat.ptr(b1), nat.ptr(s1), nat.ptr(t1), nat.ptr(w2),
                                  nat.ptr(b2), nat.ptr(s2), nat.ptr(t2), nat.ptr(y), B, h, w, st), "stem")

    report("stem conv1a+conv1b+pool @480x640 x64", timeit(fn, iters=10), 2.0 * 9 * B * h * w * (64 + 64 * 64))


def bench_attn():
    lib = nat.lib()
    st = nat.stream_ptr(DEV)
    B, K = 32, 1024
    R = 2 * B * K
    qkv = torch.randn((R, 768), device=DEV)
    o = torch.empty((R, 256), device=DEV)
    self_p = torch.tensor([[i * K, K, i * K, K] for i in range(2 * B)], dtype=torch.int32, device=DEV)
    cross_p = torch.tensor([[i * K, K, (B + i) * K, K] for i in range(B)]
                           + [[(B + i) * K, K, i * K, K] for i in range(B)], dtype=torch.int32, device=DEV)
    for name, pt in (("self attention 64 x (1024x1024), 4 heads", self_p), ("cross attention (2 directions)", cross_p)):
        def fn():
            nat.check(lib.gfc_attention(nat.ptr(qkv), 768, nat.ptr(qkv[:, 256:]) if False else
                                        nat.c_void_p(qkv.data_ptr() + 256 * 4), 768,
                                        nat.c_void_p(qkv.data_ptr() + 512 * 4), 768, nat.ptr(o), 256, nat.ptr(pt),
                                        2 * B, K, 4, 0.125, None, 0, st), "attention")

        report(name, timeit(fn), 2 * B * 4 * 2 * 2.0 * K * K * 64)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    if args.only in ("", "gemm"):
        bench_gemm()
    if args.only == "wino":
        bench_conv_wino()
        bench_conv()
        bench_stem()
    if args.only == "ffn":
        bench_ffn_fused()
    if args.only == "sweep":
        bench_gemm_sweep()
    if args.only == "msweep":
        bench_gemm_msweep()
    if args.only == "small":
        bench_gemm_small()
    if args.only in ("", "conv", "stem"):
        bench_stem()
    if args.only in ("", "conv"):
        bench_conv()
    if args.only in ("", "attn"):
        bench_attn()

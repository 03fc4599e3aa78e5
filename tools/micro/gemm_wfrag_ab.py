"""Weight-direct GEMM (weights in MFMA-fragment order, only the activation tile staged through LDS) against the staged
gemm_nt_kernel on the LightGlue shapes of the benchmark: bit-equality of the outputs and time per launch."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import _native as nat

DEV = torch.device("cuda", 0)
lib = nat.lib()
c = ctypes
lib.gfc_linear_frag_floats.restype = c.c_size_t
lib.gfc_linear_frag_floats.argtypes = [c.c_int, c.c_int]
lib.gfc_pack_linear_frag.restype = c.c_int
lib.gfc_pack_linear_frag.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_void_p]
lib.gfc_linear_wfrag.restype = c.c_int
lib.gfc_linear_wfrag.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_int, c.c_int, c.c_void_p, c.c_void_p, c.c_void_p,
                                 c.c_void_p, c.c_float, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int, c.c_void_p, c.c_int,
                                 c.c_int, c.c_int, c.c_void_p]
st = nat.stream_ptr(DEV)


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


def main():
    torch.manual_seed(0)
    for R in (65536, 16461):
        x = torch.randn((R, 256), device=DEV); msg = torch.randn((R, 256), device=DEV); h = torch.randn((R, 512), device=DEV)
        cos = torch.rand((R, 64), device=DEV); sin = torch.rand((R, 64), device=DEV)
        cases = [("qkv  N=768 K=256 rotary", x, None, 768, 256, 0, True, False), ("qkv' N=512 K=256", x, None, 512, 256, 0, False, False),
                 ("out  N=256 K=256", x, None, 256, 256, 0, False, False), ("ffn0 N=512 K=256+256", x, msg, 512, 256, 256, False, False),
                 ("ffn3 N=256 K=512 residual", h, None, 256, 512, 0, False, True), ("N=320 K=256 (ragged N)", x, None, 320, 256, 0, False, False)]
        for name, a0, a1, n, k0, k1, rot, res in cases:
            w = torch.randn((n, k0 + k1), device=DEV) / 16
            b = torch.randn((n,), device=DEV)
            resid = torch.randn((R, n), device=DEV) if res else None
            wf = torch.empty(lib.gfc_linear_frag_floats(n, k0 + k1), device=DEV)
            nat.check(lib.gfc_pack_linear_frag(nat.ptr(w), k0 + k1, n, k0 + k1, nat.ptr(wf), st), "pack")
            y0 = torch.empty((R, n), device=DEV); y1 = torch.full((R, n), float("nan"), device=DEV)
            rc, rs, rcols = (nat.ptr(cos), nat.ptr(sin), 512) if rot else (None, None, 0)

            def staged():
                nat.check(lib.gfc_linear(nat.ptr(a0), a0.shape[1], k0, nat.ptr(a1), 0 if a1 is None else a1.shape[1], k1, nat.ptr(w), k0 + k1,
                                         nat.ptr(b), None, None, 1.0, nat.ptr(resid), rc, rs, rcols, nat.ptr(y0), n, R, n, st), "linear")

            def direct():
                nat.check(lib.gfc_linear_wfrag(nat.ptr(a0), a0.shape[1], k0, nat.ptr(a1), 0 if a1 is None else a1.shape[1], k1, nat.ptr(wf),
                                               nat.ptr(b), None, None, 1.0, nat.ptr(resid), rc, rs, rcols, nat.ptr(y1), n, R, n, st), "wfrag")
            staged(); direct(); torch.cuda.synchronize()
            same = torch.equal(y0, y1)
            err = float((y0 - y1).abs().max())
            ts = [timeit(staged), timeit(direct), timeit(staged), timeit(direct)]
            fl = 2.0 * R * n * (k0 + k1)
            print(f"R={R:6d} {name:28s} identical {same} (max diff {err:.2e})  staged {min(ts[0], ts[2]) * 1e6:7.1f} us  weight-direct "
                  f"{min(ts[1], ts[3]) * 1e6:7.1f} us  ({fl / min(ts[1], ts[3]) / 1e12:6.1f} TFLOP/s)  {100 * (min(ts[1], ts[3]) / min(ts[0], ts[2]) - 1):+.1f} %", flush=True)


if __name__ == "__main__":
    main()

"""Probe: batch-1 pairs processed by N host threads, each with its own HIP stream and module instances (own
workspaces), vs one after the other.  Prints pairs/s."""
import os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import lightglue, superpoint_open, synthetic

dev = torch.device("cuda", 0)
H, W, K = 480, 640, 1024
def make():
    ext = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": K, "detection_threshold": 0.0, "nms_radius": 3,
                                      "force_num_keypoints": BATCH > 1}).eval().to(dev)
    mat = lightglue.LightGlue({"weights": "synthetic", "filter_threshold": 0.1}).eval().to(dev)
    return ext, mat
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 1        # pairs per call
THREADS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4]
pairs = [synthetic.synthetic_pairs(BATCH, H, W, seed=100 + i, device=dev) for i in range(4)]
size = torch.tensor([[float(W), float(H)]] * BATCH, device=dev)
def one(ext, mat, v0, v1):
    pj = ext({"image": torch.cat([v0, v1], 0)})
    p0 = {k: v[:BATCH] for k, v in pj.items()}
    p1 = {k: v[BATCH:] for k, v in pj.items()}
    return mat({"keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"], "descriptors0": p0["descriptors"],
                "descriptors1": p1["descriptors"], "view0": {"image_size": size}, "view1": {"image_size": size}})
def worker(n_iter, models, stream, out, tid):
    ext, mat = models
    with torch.no_grad(), torch.cuda.stream(stream):
        for i in range(n_iter):
            v0, v1 = pairs[(i + tid) % len(pairs)]
            r = one(ext, mat, v0, v1)
        stream.synchronize()
    out[tid] = int((r["matches0"] >= 0).sum())
for nthreads in THREADS:
    models = [make() for _ in range(nthreads)]
    streams = [torch.cuda.Stream(dev) for _ in range(nthreads)]
    out = {}
    for t in range(nthreads):  # warm-up
        worker(3, models[t], streams[t], out, t)
    torch.cuda.synchronize()
    n_iter = max(60 // BATCH, 6)
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(n_iter, models[t], streams[t], out, t)) for t in range(nthreads)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch {BATCH} threads {nthreads}: {BATCH * nthreads * n_iter / dt:.1f} pairs/s  (matches {out})", flush=True)

"""Batch-1 regime (HPatches-style evaluation loop): one VGA pair at a time through TwoViewPipeline.
    python tools/micro/batch1_probe.py [n_pairs]
Prints ms per pair; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_colon_amd import synthetic
from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
dev = torch.device("cuda", 0)
H, W, K = 480, 640, 1024
for joint, profiled in ((True, True), (False, True), (True, False)):
    pipe = TwoViewPipeline({
        "extractor": {"name": "extractors.superpoint_open", "weights": "synthetic", "max_num_keypoints": K,
                      "detection_threshold": 0.0, "nms_radius": 3},
        "matcher": {"name": "matchers.lightglue", "weights": "synthetic", "filter_threshold": 0.1},
        "joint_extraction": joint, "profile_calls": profiled}).eval().to(dev)
    v0, v1 = synthetic.synthetic_pairs(n, H, W, seed=4321, device=dev)
    size = torch.tensor([[float(W), float(H)]], device=dev)
    pairs = [{"view0": {"image": v0[i:i + 1], "image_size": size}, "view1": {"image": v1[i:i + 1], "image_size": size}}
             for i in range(n)]
    with torch.no_grad():
        for i in range(4):
            pred = pipe(pairs[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for d in pairs:
            pred = pipe(d)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    stages = (f", extractor {float(pred['extractor_time_ms'][0]):.3f} ms, matcher {float(pred['matcher_time_ms'][0]):.3f} ms"
              if profiled else "")
    print(f"joint_extraction={joint} profile_calls={profiled}: {dt / n * 1e3:.3f} ms per pair, "
          f"{int((pred['matches0'] >= 0).sum())} matches{stages}", flush=True)

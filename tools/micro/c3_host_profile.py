"""Where the HOST time of the pair-batched export loop goes (cProfile, main thread): resident tensors and host uint8 images.
    python tools/micro/c3_host_profile.py [n_pairs = 64]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import export_predictions as ep  # noqa: E402
from glue_factory_colon_amd import synthetic  # noqa: E402
from glue_factory_colon_amd.image_preprocessor import HostImageFeeder  # noqa: E402
from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4  # the list is walked `reps` times per pass: start-up (the first batch waits for its copies) amortised
dev = torch.device("cuda", 0)
pipe = TwoViewPipeline({
    "extractor": {"name": "gluefactory_nonfree.superpoint", "weights": "synthetic", "max_num_keypoints": 1024,
                  "detection_threshold": 0.0, "nms_radius": 3},
    "matcher": {"name": "matchers.lightglue_pretrained", "features": "superpoint", "weights": "synthetic",
                "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1},
    "profile_calls": False}).eval().to(dev)
keys = ["keypoints0", "keypoints1", "matches0", "matches1", "matching_scores0", "matching_scores1"]
optional = ["keypoint_scores0", "keypoint_scores1"]
raw = synthetic.hpatches_like_host_images(n)
conf = {"resize": 480, "side": "short"}
resident = list(HostImageFeeder(raw, conf))


def run(source):
    out = []
    ep._export_loop(enumerate(source), pipe, "cuda", keys, optional, None, False, 1, out, 32)
    return out


with torch.no_grad():
    for tag, make in (("resident", lambda: resident * reps), ("from_host", lambda: HostImageFeeder(raw * reps, conf))):
        run(make())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(make())
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"== {tag}: {1e3 * t_all / (n * reps):.3f} ms/pair wall ({n * reps / t_all:.1f} pairs/s); main thread returned "
              f"after {1e3 * t_issue / (n * reps):.3f} ms/pair")
        pr = cProfile.Profile()
        pr.enable()
        run(make())
        pr.disable()
        torch.cuda.synchronize()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
        print("\n".join(s.getvalue().splitlines()[:40]))

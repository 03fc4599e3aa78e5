"""Where does a pair-batched evaluation batch spend its time?  64 HPatches-shaped pairs (five image shapes) through
export_predictions' loop at pair_batch = N, stage by stage (device-synchronised): extractor calls per shape, the ragged
matcher, record packing + host copy.   python tools/micro/c3_regime_probe.py [pair_batch]
(under `rocprofv3 --kernel-trace --stats` for the per-kernel split)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import export_predictions as ep  # noqa: E402
from glue_factory_colon_amd import synthetic  # noqa: E402
from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline  # noqa: E402

pb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
items = synthetic.hpatches_shaped_pairs(64, device=dev)
pipe = TwoViewPipeline({
    "extractor": {"name": "gluefactory_nonfree.superpoint", "weights": "synthetic", "max_num_keypoints": 1024,
                  "detection_threshold": 0.0, "nms_radius": 3},
    "matcher": {"name": "matchers.lightglue_pretrained", "features": "superpoint", "weights": "synthetic",
                "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1},
    "profile_calls": False}).eval().to(dev)
keys = ["keypoints0", "keypoints1", "matches0", "matches1", "matching_scores0", "matching_scores1"]
opt = ["keypoint_scores0", "keypoint_scores1"]


def sync():
    torch.cuda.synchronize(dev)
    return time.perf_counter()


with torch.no_grad():
    for _ in range(2):
        out = []
        ep._export_loop(enumerate(items), pipe, "cuda", keys, opt, None, False, 1, out, pb)
    t0 = sync()
    out = []
    ep._export_loop(enumerate(items), pipe, "cuda", keys, opt, None, False, 1, out, pb)
    t1 = sync()
    print(f"pair_batch {pb}: whole loop {64 / (t1 - t0):.1f} pairs/s ({(t1 - t0) / 64 * 1e3:.3f} ms per pair)")
    # stage by stage on the first batch
    datas = items[:pb]
    views = [d[f"view{i}"] for d in datas for i in ("0", "1")]
    ext, mat = pipe.extractor, pipe.matcher
    for _ in range(2):
        vp = ext.forward_views(views)
    t0 = sync()
    for _ in range(5):
        vp = ext.forward_views(views)
    t1 = sync()
    shapes = {}
    for v in views:
        shapes[tuple(v["image"].shape[-2:])] = shapes.get(tuple(v["image"].shape[-2:]), 0) + 1
    print(f"  extractor, {len(views)} views in {len(shapes)} shape groups {sorted(shapes.items())}: {(t1 - t0) / 5 * 1e3:.2f} ms per batch")
    pair_items = []
    for j, d in enumerate(datas):
        p0, p1 = vp[2 * j], vp[2 * j + 1]
        pair_items.append({**d, **{k + "0": v for k, v in p0.items()}, **{k + "1": v for k, v in p1.items()}})
    for _ in range(2):
        mo = mat.forward_pairs(pair_items)
    t0 = sync()
    for _ in range(5):
        mo = mat.forward_pairs(pair_items)
    t1 = sync()
    print(f"  ragged matcher over {len(pair_items)} pairs: {(t1 - t0) / 5 * 1e3:.2f} ms per batch")
    t0 = sync()
    for _ in range(5):
        preds = pipe.forward_pairs(datas)
    t1 = sync()
    print(f"  TwoViewPipeline.forward_pairs: {(t1 - t0) / 5 * 1e3:.2f} ms per batch")
    t0 = sync()
    for _ in range(5):
        recs = ep._process_batch(pipe, datas, keys, opt, None, False)
    t1 = sync()
    print(f"  _process_batch (forward + records + host copy): {(t1 - t0) / 5 * 1e3:.2f} ms per batch")

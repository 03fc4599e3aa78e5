"""HPatches evaluation on the GPU path -- counterpart of `gluefactory.eval.hpatches.HPatchesPipeline`
(reference gluefactory/eval/hpatches.py:28-199) for BASELINE config 3: a local `hpatches-sequences-release` directory
-> files decoded on the host (`hpatches.HPatches`, `image_io`) -> images preprocessed on the GPU (`HostImageFeeder`) ->
`export_predictions` (pair batches, a sequence's reference image extracted once, pairs sharded over the ranks with one
gather) -> `predictions.h5` in the reference's layout -> per-pair match metrics and DLT homography error on the GPU
(`eval_utils`) -> the reference's summaries (`med_*`, `mean_*`, `H_error_dlt@{1,3,5}px`).

    python -m glue_factory_colon_amd.eval_hpatches --data_dir /data/hpatches-sequences-release \\
        --extractor_weights superpoint_v6_from_tf.pth --matcher_weights superpoint_lightglue.pth [--gpus 8]

What is NOT here: the robust estimators of `eval_homography_robust` (eval/hpatches.py:146-152: poselib / OpenCV RANSAC,
external CPU libraries absent from this image) and with them `H_error_ransac@*`; figures.  Everything that is here keeps
the reference's names, so a results table lines up key by key.
"""
import argparse
import json
import os
import sys
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

from . import eval_utils, hpatches
from .cache_loader import CacheLoader
from .export_predictions import export_predictions

EXPORT_KEYS = ["keypoints0", "keypoints1", "matches0", "matches1", "matching_scores0", "matching_scores1"]
TIMING_KEYS = ["extractor_time_ms", "extractor_core_time_ms", "matcher_time_ms", "total_time_ms"]
MEMORY_KEYS = ["extractor_memory_mb", "matcher_memory_mb", "forward_allocated_memory_mb", "forward_reserved_memory_mb"]
CONTEXT_KEYS = ["pair_resolution"]
OPTIONAL_EXPORT_KEYS = ["keypoint_scores0", "keypoint_scores1", *TIMING_KEYS, *MEMORY_KEYS, *CONTEXT_KEYS]
DEFAULT_DATA_CONF = {"data_dir": "hpatches-sequences-release", "preprocessing": {"resize": 480, "side": "short"}}


def cal_error_auc(errors, thresholds):
    """Area under the recall-vs-error curve up to each threshold, normalised (gluefactory/utils/tools.py:137-149)."""
    errors = np.sort(np.asarray(errors, dtype=np.float64))
    recall = (np.arange(len(errors)) + 1) / len(errors)
    errors = np.r_[0.0, errors]
    recall = np.r_[0.0, recall]
    trapz = getattr(np, "trapezoid", None) or np.trapz
    aucs = []
    for t in thresholds:
        last = np.searchsorted(errors, t)
        r = np.r_[recall[:last], recall[last - 1]]
        e = np.r_[errors[:last], t]
        aucs.append(np.round(trapz(r, x=e) / t, 4))
    return aucs


class HPatchesPipeline:
    export_keys = EXPORT_KEYS
    optional_export_keys = OPTIONAL_EXPORT_KEYS

    def __init__(self, data_conf=None, pair_batch=32):
        self.data_conf = {**DEFAULT_DATA_CONF, **dict(data_conf or {})}
        self.pair_batch = int(pair_batch)
        self.dataset = hpatches.HPatches(self.data_conf)

    def get_predictions(self, experiment_dir, model, overwrite=False):
        """eval/hpatches.py:98-110.  Under torch.distributed every rank calls this; rank 0 writes the file."""
        pred_file = Path(experiment_dir) / "predictions.h5"
        if not pred_file.exists() or overwrite:
            export_predictions(self.dataset.feeder(), model, pred_file, keys=self.export_keys,
                               optional_keys=self.optional_export_keys, pair_batch=self.pair_batch,
                               view_key=self.dataset.view_key if self.pair_batch > 1 else None, shard_group=5)
        return pred_file

    def run_eval(self, pred_file, device="cuda"):
        """eval/hpatches.py:112-176 without the robust estimators -> (summaries, results)."""
        pred_file = Path(pred_file)
        assert pred_file.exists()
        cache = CacheLoader({"path": str(pred_file), "collate": None, "add_data_path": False, "device": str(device)}).eval()
        results = defaultdict(list)
        for i in range(len(self.dataset)):
            data = self.dataset.meta(i)
            on_dev = {"name": [data["name"]], "view0": {"scales": data["view0"]["scales"][None].to(device)},
                      "view1": {"scales": data["view1"]["scales"][None].to(device)}}
            pred = cache(on_dev)  # key points back in the coordinates of the preprocessed images (x scales)
            ev = {"H_0to1": data["H_0to1"].to(device), "view0": {"image_size": data["view0"]["image_size"].to(device)}}
            results_i = {}
            if "keypoints0" in pred:
                results_i = {**eval_utils.eval_matches_homography(ev, pred), **eval_utils.eval_homography_dlt(ev, pred)}
            for k in (*TIMING_KEYS, *MEMORY_KEYS, *CONTEXT_KEYS):
                if k in pred:
                    results_i[k] = pred[k].item()
            results_i["names"] = data["name"]
            results_i["scenes"] = data["scene"]
            for k, v in results_i.items():
                results[k].append(v)
        summaries = {}
        for k, v in results.items():
            arr = np.array(v)
            if not np.issubdtype(arr.dtype, np.number):
                continue
            summaries[f"med_{k}"] = round(float(np.median(arr)), 3)
            summaries[f"mean_{k}"] = round(float(np.mean(arr)), 3)
        if "H_error_dlt" in results:
            for th, auc in zip([1, 3, 5], cal_error_auc(results["H_error_dlt"], [1, 3, 5])):
                summaries[f"H_error_dlt@{th}px"] = float(auc)
        return summaries, dict(results)

    def run(self, experiment_dir, model, overwrite=False):
        """Predictions (all ranks), then the evaluation on rank 0.  Returns (summaries, results) on rank 0, (None, None)
        elsewhere."""
        import torch.distributed as dist

        pred_file = self.get_predictions(experiment_dir, model, overwrite=overwrite)
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return None, None
        return self.run_eval(pred_file)


def build_model(extractor_weights, matcher_weights, official=True, max_num_keypoints=1024):
    """The `superpoint+lightglue-official` configuration (gluefactory/configs/superpoint+lightglue-official.yaml) on this
    package's modules; official=False: `superpoint-open+lightglue`."""
    from .two_view_pipeline import TwoViewPipeline

    if official:
        conf = {"extractor": {"name": "gluefactory_nonfree.superpoint", "weights": extractor_weights,
                              "max_num_keypoints": max_num_keypoints, "detection_threshold": 0.0, "nms_radius": 3},
                "matcher": {"name": "matchers.lightglue_pretrained", "features": "superpoint", "weights": matcher_weights,
                            "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}}
    else:
        conf = {"extractor": {"name": "extractors.superpoint_open", "weights": extractor_weights,
                              "max_num_keypoints": max_num_keypoints, "detection_threshold": 0.0, "nms_radius": 3},
                "matcher": {"name": "matchers.lightglue", "weights": matcher_weights, "filter_threshold": 0.1,
                            "depth_confidence": -1, "width_confidence": -1}}
    return TwoViewPipeline(conf).eval()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--data_dir", required=True)
    ap.add_argument("--experiment_dir", default="outputs/hpatches")
    ap.add_argument("--extractor_weights", default="synthetic", help="local .pth (reference key names) or 'synthetic'")
    ap.add_argument("--matcher_weights", default="synthetic")
    ap.add_argument("--open", action="store_true", help="superpoint-open + in-tree lightglue instead of the official pair")
    ap.add_argument("--subset", default=None, choices=[None, "i", "v"])
    ap.add_argument("--pair_batch", type=int, default=32)
    ap.add_argument("--overwrite", action="store_true")
    ap.add_argument("--gpus", type=int, default=1, help="> 1 without a launcher: this process starts the ranks itself")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"), "-m",
               "glue_factory_colon_amd.eval_hpatches", *(argv if argv is not None else sys.argv[1:])]
        return subprocess.call(cmd)  # (no GPU call was made by this process)
    from . import sharding

    rank, world, local = sharding.init_from_env("nccl") if "WORLD_SIZE" in os.environ else (0, 1, 0)
    torch.cuda.set_device(local)
    pipe = HPatchesPipeline({"data_dir": args.data_dir, "subset": args.subset}, pair_batch=args.pair_batch)
    model = build_model(args.extractor_weights, args.matcher_weights, official=not args.open).to(f"cuda:{local}")
    summaries, _ = pipe.run(args.experiment_dir, model, overwrite=args.overwrite)
    if rank == 0:
        print(json.dumps(summaries, indent=1))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

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

    def __init__(self, data_conf=None, pair_batch=32, num_workers=2):
        self.data_conf = {**DEFAULT_DATA_CONF, **dict(data_conf or {})}
        self.pair_batch, self.num_workers = int(pair_batch), int(num_workers)
        self.dataset = hpatches.HPatches(self.data_conf)

    def get_predictions(self, experiment_dir, model, overwrite=False):
        """eval/hpatches.py:98-110.  Under torch.distributed every rank calls this; rank 0 writes the file."""
        pred_file = Path(experiment_dir) / "predictions.h5"
        if not pred_file.exists() or overwrite:
            export_predictions(self.dataset.feeder(num_workers=self.num_workers, depth=max(64, 2 * self.pair_batch)), model, pred_file, keys=self.export_keys,
                               optional_keys=self.optional_export_keys, pair_batch=self.pair_batch,
                               view_key=self.dataset.view_key if self.pair_batch > 1 else None, shard_group=5)
        return pred_file

    def run_eval(self, pred_file, device="cuda"):
        """eval/hpatches.py:112-176 without the robust estimators -> (summaries, results).  The reference walks the list
        pair by pair (CacheLoader -> eval_matches_homography -> eval_homography_dlt); the same per-pair arithmetic runs
        here for all pairs with equal key-point counts in ONE call of each kernel (one workgroup per pair either way:
        the results do not depend on the grouping), with the cached key points put back into the coordinates of the
        preprocessed images exactly as CacheLoader does (float32 `keypoints * scales`, cache_loader.py)."""
        from .export_predictions import load_predictions

        pred_file = Path(pred_file)
        assert pred_file.exists()
        records = load_predictions(pred_file)
        metas = [self.dataset.meta(i) for i in range(len(self.dataset))]
        per_pair = [None] * len(metas)
        groups = defaultdict(list)
        for i, meta in enumerate(metas):
            rec = records[meta["name"]]
            per_pair[i] = {k: rec[k].item() for k in (*TIMING_KEYS, *MEMORY_KEYS, *CONTEXT_KEYS) if k in rec}
            if "keypoints0" in rec:
                groups[(rec["keypoints0"].shape[0], rec["keypoints1"].shape[0])].append(i)
        for idxs in groups.values():
            recs = [records[metas[i]["name"]] for i in idxs]

            def stack(key, dtype):
                return torch.from_numpy(np.stack([r[key] for r in recs])).to(device=device, dtype=dtype)

            sc0 = torch.stack([metas[i]["view0"]["scales"] for i in idxs]).to(device)[:, None]
            sc1 = torch.stack([metas[i]["view1"]["scales"] for i in idxs]).to(device)[:, None]
            kp0, kp1 = stack("keypoints0", torch.float32) * sc0, stack("keypoints1", torch.float32) * sc1
            m0, s0 = stack("matches0", torch.long), stack("matching_scores0", torch.float32)
            H = torch.stack([metas[i]["H_0to1"] for i in idxs]).to(device)
            size0 = torch.stack([metas[i]["view0"]["image_size"] for i in idxs]).to(device)
            metrics = eval_utils.match_metrics(H, kp0, kp1, m0).cpu()
            _, err = eval_utils.homography_dlt(H, kp0, kp1, m0, s0, size0)
            err = err.cpu()
            for j, i in enumerate(idxs):
                for c, key in enumerate(eval_utils.RESULT_KEYS):
                    v = metrics[j, c].item()
                    per_pair[i][key] = int(v) if key == "num_matches" else float(v)
                per_pair[i]["H_error_dlt"] = float(err[j])
        results = defaultdict(list)
        for i, meta in enumerate(metas):
            results_i = {k: per_pair[i][k] for k in (*eval_utils.RESULT_KEYS, "H_error_dlt") if k in per_pair[i]}
            results_i.update({k: v for k, v in per_pair[i].items() if k not in results_i})
            results_i["names"] = meta["name"]
            results_i["scenes"] = meta["scene"]
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

    def run_eval_pairwise(self, pred_file, device="cuda"):
        """The reference's own loop shape (one pair at a time through CacheLoader and the drop-in `eval_*` functions):
        kept as the check of `run_eval` (tests/test_hpatches_reader.py)."""
        cache = CacheLoader({"path": str(pred_file), "collate": None, "add_data_path": False, "device": str(device)}).eval()
        results = defaultdict(list)
        for i in range(len(self.dataset)):
            data = self.dataset.meta(i)
            on_dev = {"name": [data["name"]], "view0": {"scales": data["view0"]["scales"][None].to(device)},
                      "view1": {"scales": data["view1"]["scales"][None].to(device)}}
            pred = cache(on_dev)
            ev = {"H_0to1": data["H_0to1"].to(device), "view0": {"image_size": data["view0"]["image_size"].to(device)}}
            results_i = {**eval_utils.eval_matches_homography(ev, pred), **eval_utils.eval_homography_dlt(ev, pred)}
            results_i["names"] = data["name"]
            for k, v in results_i.items():
                results[k].append(v)
        return dict(results)

    def run(self, experiment_dir, model, overwrite=False):
        """Predictions (all ranks), then the evaluation on rank 0.  Returns (summaries, results) on rank 0, (None, None)
        elsewhere."""
        import torch.distributed as dist

        pred_file = self.get_predictions(experiment_dir, model, overwrite=overwrite)
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return None, None
        return self.run_eval(pred_file)


def build_model(extractor_weights, matcher_weights, official=True, max_num_keypoints=1024, profile_calls=False):
    """The `superpoint+lightglue-official` configuration (gluefactory/configs/superpoint+lightglue-official.yaml) on this
    package's modules; official=False: `superpoint-open+lightglue`.  profile_calls=True restores the reference's device
    synchronisations around every extractor / matcher call and with them the optional timing / memory keys of the records
    (two_view_pipeline.py:78-102); without them the host prepares the next pair batch while the GPU works on this one."""
    from .two_view_pipeline import TwoViewPipeline

    if official:
        conf = {"extractor": {"name": "gluefactory_nonfree.superpoint", "weights": extractor_weights,
                              "max_num_keypoints": max_num_keypoints, "detection_threshold": 0.0, "nms_radius": 3},
                "matcher": {"name": "matchers.lightglue_pretrained", "features": "superpoint", "weights": matcher_weights,
                            "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1},
                "profile_calls": bool(profile_calls)}
    else:
        conf = {"extractor": {"name": "extractors.superpoint_open", "weights": extractor_weights,
                              "max_num_keypoints": max_num_keypoints, "detection_threshold": 0.0, "nms_radius": 3},
                "matcher": {"name": "matchers.lightglue", "weights": matcher_weights, "filter_threshold": 0.1,
                            "depth_confidence": -1, "width_confidence": -1},
                "profile_calls": bool(profile_calls)}
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
    ap.add_argument("--num_workers", type=int, default=2, help="reader threads that read the image files ahead of the GPU")
    ap.add_argument("--overwrite", action="store_true")
    ap.add_argument("--profile_calls", action="store_true",
                    help="the reference's per-call device synchronisations and timing keys (slower: no host / GPU overlap)")
    ap.add_argument("--gpus", type=int, default=1, help="> 1 without a launcher: this process starts the ranks itself")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess

        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"), "-m",
               "glue_factory_colon_amd.eval_hpatches", *(argv if argv is not None else sys.argv[1:])]
        return subprocess.call(cmd)  # (no GPU call was made by this process)
    from . import sharding

    # torch starts as many CPU threads as the host has cores (256 on an MI355X node); this process only does small host
    # tensor operations, which a pool of that size slows down 4-10x on a per-GPU share of the cores
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 16)))
    rank, world, local = sharding.init_from_env("nccl") if "WORLD_SIZE" in os.environ else (0, 1, 0)
    torch.cuda.set_device(local)
    pipe = HPatchesPipeline({"data_dir": args.data_dir, "subset": args.subset}, pair_batch=args.pair_batch,
                            num_workers=args.num_workers)
    model = build_model(args.extractor_weights, args.matcher_weights, official=not args.open,
                        profile_calls=args.profile_calls).to(f"cuda:{local}")
    summaries, _ = pipe.run(args.experiment_dir, model, overwrite=args.overwrite)
    if rank == 0:
        print(json.dumps(summaries, indent=1))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

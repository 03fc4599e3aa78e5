#!/usr/bin/env python3
"""Benchmark of the SuperPoint + LightGlue hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
     (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or as the plain command above: the parent then
     starts the N rank processes itself -- before making any GPU call -- and relays rank 0's JSON line)

Metric (BASELINE.json): image-pairs/sec, SuperPoint + LightGlue, 1024 keypoints, 640x480.
One "step" = one pass of the hot path over one batch of synthetic pairs resident in HBM:
extractor on view0 and view1, then the matcher -- the work of TwoViewPipeline._forward (reference
gluefactory/models/two_view_pipeline.py:278-339) without its per-call device syncs; by default both views'
images go through one extractor call of 2*pairs images (--joint-extract 0 = two calls, same results).
Every rank processes its own `--pairs` image pairs per step (weak scaling, no data-path
collective); one RCCL gather of per-pair records closes the job.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = the one with the largest share of the step:
`attention_kernel<2,4>`, 18 launches per step, timed live with HIP events around each launch inside the timed
region; `frac` = ALGORITHMIC attention FLOPs / launch time / fp32-MFMA peak, never above 1; the stem convolution
and the split by self / cross launch are listed under `roofline.kernels`) and `cpu_baseline` (the CPU oracle, a
PyTorch-CPU port of the reference path, on a bounded sample; also on the N > 1 line, run by rank 0 after the
timed region and the gather).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from glue_factory_colon_amd import _native as nat  # noqa: E402
from glue_factory_colon_amd import lightglue, sharding, superpoint_open, synthetic, weights  # noqa: E402

METRIC = "image-pairs/sec (SuperPoint+LightGlue, 1024 kpts, 640x480)"
H, W, K = 480, 640, 1024
# algorithmic FLOPs (2*MAC) of the stem kernel per image: conv1a 1->64 and conv1b 64->64 @480x640
# (0.35 + 22.65 GFLOP; the halo recomputation of conv1a is not counted)
STEM_FLOPS_PER_IMAGE = 2 * 9 * (1 * 64 + 64 * 64) * 480 * 640
PAIR_FLOPS = 182.3e9  # whole path per pair at this configuration (direct arithmetic)
# what the matrix pipe executes per pair on the default path: the stem (conv1a + conv1b) as 2504 MFMAs per
# 32x16-pixel item = 6.15 GFLOP / image (F(4x4,3x3)), the other 3x3 convolutions (28.31 GFLOP direct) at 4/9 (Winograd
# F(2x2,3x3)), 1x1 heads and LightGlue as counted by SURVEY.md 8d with out_proj / to_out folded into ffn[0]
# (-9 x 4 x 2*1024*256*256 = -4.83 GFLOP) and the second cross direction's sim re-evaluated (+9 x 2*1024*1024*64*4 =
# +4.83 GFLOP): 2 x (6.15 + 4/9 x 28.31 + 0.79) + 78.11 = 117.2 GFLOP
PAIR_FLOPS_EXECUTED = 2 * (2504 * 4096 * 600 + 4.0 / 9.0 * 28.31e9 + 0.79e9) + 78.11e9 - 4.83e9 + 4.83e9
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


PMC_SUMMARIES = ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json")  # newest first


def pmc_traffic_bytes(kernel_prefix):
    """(HBM bytes per launch, file) of a kernel from the committed rocprofv3 --pmc passes of this same default
    command (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes; tools/pmc_summary.py, tools/profile_r06.sh).
    PMC counters cannot be collected from inside the process, so the figure is read from profiles/; (None, None)
    if absent."""
    for name in PMC_SUMMARIES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                summary = json.load(f)
            key = next(k for k in summary if k.replace("void ", "").startswith(kernel_prefix))
            return summary[key]["hbm_bytes_per_launch"], "profiles/" + name
        except (OSError, KeyError, ValueError, StopIteration):
            continue
    return None, None


def cpu_baseline(n_pairs: int, iters: int):
    """The oracle (PyTorch-CPU port of the reference path) on the same kind of input, host cores."""
    from oracle import lightglue as olg
    from oracle import superpoint as osp

    v0, v1 = synthetic.synthetic_pairs(n_pairs, H, W, seed=1234)
    sd_sp, sd_lg = weights.superpoint_open_state_dict(0), weights.lightglue_state_dict(0)
    size = torch.tensor([[float(W), float(H)]] * n_pairs)

    def run():
        a = osp.extract(sd_sp, v0, "open", nms_radius=3, max_num_keypoints=K, detection_threshold=0.0)
        b = osp.extract(sd_sp, v1, "open", nms_radius=3, max_num_keypoints=K, detection_threshold=0.0)
        out = olg.match(sd_lg, torch.stack(a["keypoints"]), torch.stack(b["keypoints"]),
                        torch.stack(a["descriptors"]), torch.stack(b["descriptors"]), size, size,
                        filter_threshold=0.1)
        return int((out["matches0"] >= 0).sum())

    # pick the intra-op thread count that serves this small-batch workload best on this host
    # (the default = all logical CPUs oversubscribes badly on a 256-thread box)
    default_threads = torch.get_num_threads()
    probe = {}
    for nt in sorted({8, 16, 32, 64, default_threads}):
        if nt > default_threads:
            continue
        torch.set_num_threads(nt)
        run()  # warm-up at this setting
        t0 = time.perf_counter()
        run()
        probe[nt] = time.perf_counter() - t0
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return {"value": round(n_pairs * iters / dt, 4), "unit": "image-pairs/sec", "cores": best, "kind": "port",
            "sample": f"{iters} iterations of {n_pairs} VGA pairs, 1024 kpts, oracle (PyTorch-CPU fp32 port of the "
                      f"reference path); {best} torch threads (best of {sorted(probe)}; host has {os.cpu_count()} "
                      f"logical CPUs), {dt:.1f} s timed"}


def torch_eager_same_gpu(dev, n_pairs: int = 16, iters: int = 4):
    """Part of the baseline leg: the same oracle with its tensors on the MI355X, i.e. what the reference's own
    PyTorch modules do on this GPU under PyTorch-ROCm eager fp32 (MIOpen convolutions, rocBLAS linears, SDPA).
    Reported for context beside `cpu_baseline`; never the thing shipped."""
    from oracle import lightglue as olg
    from oracle import superpoint as osp

    v0, v1 = synthetic.synthetic_pairs(n_pairs, H, W, seed=1234, device=dev)
    sd_sp = {k: v.to(dev) for k, v in weights.superpoint_open_state_dict(0).items()}
    sd_lg = {k: v.to(dev) for k, v in weights.lightglue_state_dict(0).items()}
    size = torch.tensor([[float(W), float(H)]] * n_pairs, device=dev)

    def run():
        a = osp.extract(sd_sp, v0, "open", nms_radius=3, max_num_keypoints=K, detection_threshold=0.0)
        b = osp.extract(sd_sp, v1, "open", nms_radius=3, max_num_keypoints=K, detection_threshold=0.0)
        return olg.match(sd_lg, torch.stack(a["keypoints"]), torch.stack(b["keypoints"]),
                         torch.stack(a["descriptors"]), torch.stack(b["descriptors"]), size, size, filter_threshold=0.1)

    with torch.no_grad():
        run()
        run()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    return {"value": round(n_pairs * iters / dt, 2), "unit": "image-pairs/sec",
            "sample": f"{iters} iterations of {n_pairs} VGA pairs, oracle tensors on cuda:0 (PyTorch-ROCm eager fp32)"}


def batch1_latency(dev, n_pairs: int = 24, warmup: int = 4):
    """Information only (never `value`): the batch-1 regime of the HPatches evaluation loop (config 3: image sizes differ,
    so pairs cannot be batched; datasets/hpatches.py:60) -- one VGA pair at a time through TwoViewPipeline on one
    stream, with the reference's per-call profiling (device-synchronised timing of extractor and matcher,
    two_view_pipeline.py:78-102) and without it (`profile_calls: false`)."""
    from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline

    v0, v1 = synthetic.synthetic_pairs(n_pairs, H, W, seed=4321, device=dev)
    size = torch.tensor([[float(W), float(H)]], device=dev)
    pairs = [{"view0": {"image": v0[i:i + 1], "image_size": size}, "view1": {"image": v1[i:i + 1], "image_size": size}}
             for i in range(n_pairs)]
    res, spread = {}, {}
    for profiled in (True, False):
        pipe = TwoViewPipeline({
            "extractor": {"name": "extractors.superpoint_open", "weights": "synthetic", "max_num_keypoints": K,
                          "detection_threshold": 0.0, "nms_radius": 3, "force_num_keypoints": True},
            "matcher": {"name": "matchers.lightglue", "weights": "synthetic", "filter_threshold": 0.1},
            "profile_calls": profiled}).eval().to(dev)
        with torch.no_grad():
            for i in range(max(warmup, n_pairs)):  # every pair once: the caching allocator has seen each pair's sizes
                pipe(pairs[i % n_pairs])
            reps = []
            for _ in range(3):  # best of three passes: one host hiccup in a 60 ms pass would move the mean by 50 %
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for d in pairs:
                    pred = pipe(d)
                torch.cuda.synchronize(dev)
                reps.append((time.perf_counter() - t0) / n_pairs)
            res[profiled] = min(reps)
            spread[profiled] = max(reps)
    return {"pairs_per_s": round(1.0 / res[True], 1), "ms_per_pair": round(res[True] * 1e3, 3), "workers": 1,
            "ms_per_pair_unprofiled": round(res[False] * 1e3, 3), "pairs_per_s_unprofiled": round(1.0 / res[False], 1),
            "ms_per_pair_worst_pass": round(spread[True] * 1e3, 3), "ms_per_pair_unprofiled_worst_pass": round(spread[False] * 1e3, 3),
            "matches_last_pair": int((pred["matches0"] >= 0).sum()),
            "sample": f"{n_pairs} VGA pairs, 1024 kpts, one at a time through TwoViewPipeline on one stream (both views in "
                      "one extractor call), every pair run once untimed, then the best of three timed passes (the worst is "
                      "reported beside it); `unprofiled` = without the reference's per-call device synchronisations"}



def c3_regime(dev, n_pairs: int = 128, n_host: int = 64):
    """Information only (never `value`): BASELINE config 3's regime -- the HPatches evaluation loop
    (utils/export_predictions.py:36-85: batch 1 because the IMAGES differ in size; official SuperPoint + LightGlue,
    1024 key points, detection threshold 0) on an HPatches-shaped list of mixed image shapes, through this package's
    export loop without the file write: forward, key filtering, un-scaling, host copy of every record.  Sequential
    loop, `workers` (pairs in flight on several streams) and `pair_batch` (N consecutive pairs: the extractor once per
    distinct image shape, the matcher once over all N pairs with their own key-point counts).  `legs` carries, per
    leg, the match total and whether every integer output equals the first (sequential) leg's, element-wise."""
    from glue_factory_colon_amd import export_predictions as ep
    from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline

    items = synthetic.hpatches_shaped_pairs(n_pairs, device=dev)
    keys = ["keypoints0", "keypoints1", "matches0", "matches1", "matching_scores0", "matching_scores1"]
    optional = ["keypoint_scores0", "keypoint_scores1"]

    def pipeline(profiled):
        return TwoViewPipeline({
            "extractor": {"name": "gluefactory_nonfree.superpoint", "weights": "synthetic", "max_num_keypoints": K,
                          "detection_threshold": 0.0, "nms_radius": 3},
            "matcher": {"name": "matchers.lightglue_pretrained", "features": "superpoint", "weights": "synthetic",
                        "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1},
            "profile_calls": profiled}).eval().to(dev)

    def run(pipe, workers, pair_batch, source=None):
        out = []
        ep._export_loop(enumerate(items if source is None else source), pipe, "cuda", keys, optional, None, False,
                        workers, out, pair_batch)
        return out

    def compare(out, base):
        """Integer outputs of a leg's records against the sequential leg's, pair by pair (element-wise).  For a pair
        that differs: the entries that changed and how far their matching score is from filter_threshold in either
        run -- a near-tie of the threshold (floats depend on the batch size through the summation order) or not."""
        a = {name: rec for _, name, rec in base}
        diff_pairs, flipped, nearest = 0, 0, None
        for _, name, rec in out:
            ref = a[name]
            bad = [k for k in ("matches0", "matches1") if ref[k].shape != rec[k].shape or (ref[k] != rec[k]).any()]
            kp_same = all(ref[k].shape == rec[k].shape and (ref[k] == rec[k]).all() for k in ("keypoints0", "keypoints1"))
            if not bad and kp_same:
                continue
            diff_pairs += 1
            for k in bad:
                if ref[k].shape != rec[k].shape:
                    continue
                idx = (ref[k] != rec[k]).nonzero()[0]
                flipped += len(idx)
                sk = "matching_scores" + k[-1]
                for sc in (ref[sk][idx], rec[sk][idx]):
                    live = sc[sc > 0]
                    if len(live):
                        d = float(abs(live - 0.1).min())
                        nearest = d if nearest is None else min(nearest, d)
        return {"pairs_differing": diff_pairs, "entries_flipped": flipped, "min_score_distance_to_threshold": nearest}

    res, legs = {}, {}
    base = None
    pipe = pipeline(False)
    with torch.no_grad():
        for tag, profiled, workers, pb in (("sequential_reference_syncs", True, 1, 1), ("sequential", False, 1, 1),
                                           ("workers4", False, 4, 1), ("pair_batch16", False, 1, 16),
                                           ("pair_batch32", False, 1, 32)):
            p = pipeline(True) if profiled else pipe
            run(p, workers, pb)  # untimed: allocator and per-shape workspaces warm
            best = None
            for _ in range(2):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                out = run(p, workers, pb)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            res[tag] = round(n_pairs / best, 1)
            out = sorted(out, key=lambda e: e[0])
            base = out if base is None else base
            legs[tag] = {"matches_total": sum(int((rec["matches0"] >= 0).sum()) for _, _, rec in out),
                         "integers_equal_to_first_leg": None, **compare(out, base)}
            legs[tag]["integers_equal_to_first_leg"] = legs[tag]["pairs_differing"] == 0
        # a NEVER-RUN pipeline whose first call is the pair-batched loop (weights packed inside forward_views)
        fresh = run(pipeline(False), 1, 32)
        legs["pair_batch32_first_call_of_fresh_pipeline"] = {
            "matches_total": sum(int((rec["matches0"] >= 0).sum()) for _, _, rec in fresh),
            **compare(sorted(fresh, key=lambda e: e[0]), base)}
        legs["pair_batch32_first_call_of_fresh_pipeline"]["integers_equal_to_first_leg"] = \
            legs["pair_batch32_first_call_of_fresh_pipeline"]["pairs_differing"] == 0
        # the list's real structure: the five pairs of an HPatches sequence share their view 0 (datasets/hpatches.py:98-99);
        # with `view_key` the pair-batched loop extracts a shared image once per batch
        seq_items = synthetic.hpatches_shaped_pairs(n_pairs, device=dev, shared_view0=True)
        seqs = {}
        for tag, vk in (("pair_batch32", None),
                        ("pair_batch32_view_dedupe", lambda item, i: (item["scene"][0], 1) if i == 0 else None)):
            def run_seq():
                out = []
                ep._export_loop(enumerate(seq_items), pipe, "cuda", keys, optional, None, False, 1, out, 32, vk)
                return out
            run_seq()
            best = None
            for _ in range(2):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                out = run_seq()
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            seqs[tag] = {"pairs_per_s": round(n_pairs / best, 1),
                         "matches_total": sum(int((rec["matches0"] >= 0).sum()) for _, _, rec in out),
                         "_out": sorted(out, key=lambda e: e[0])}
        seqs["integers_equal"] = compare(seqs["pair_batch32_view_dedupe"].pop("_out"), seqs["pair_batch32"].pop("_out"))["pairs_differing"] == 0
        seqs["sample"] = (f"{n_pairs} pairs in sequences of five that share their view-0 image (the structure of the HPatches list), "
                          "same pipeline; `view_dedupe` = export_predictions(view_key=...) extracts a shared image once per pair batch")
        res["hpatches_sequences"] = seqs
        # the loop that FEEDS the path (datasets/hpatches.py:94-112 + utils/image.py:33-72): decoded uint8 images in
        # pinned host memory -> async H2D on a copy stream -> gfc_preprocess_resize (short side 480, antialias) ->
        # forward_pairs(32) -> records; beside it the same images already preprocessed and resident in HBM
        from glue_factory_colon_amd.image_preprocessor import HostImageFeeder
        # (the 64 distinct pairs are walked four times per pass: the first batch of a pass waits for its own copies, a
        # start-up cost that a 540-pair list amortises and a two-batch pass would not)
        raw = synthetic.hpatches_like_host_images(n_host, shared_view0=True) * 4
        n_host *= 4
        pconf = {"resize": 480, "side": "short"}
        raw_key = lambda r, i: (r["scene"], 1) if i == 0 else None      # noqa: E731  (names on the RAW items: strings)
        item_key = lambda d, i: (d["scene"][0], 1) if i == 0 else None  # noqa: E731  (on loader items: lists of one)
        host = {}
        for tag in ("from_host_uint8_pair_batch32", "from_host_uint8_pair_batch32_view_dedupe", "resident_same_images_pair_batch32"):
            best, feeder = None, None
            dedupe = tag.endswith("view_dedupe")
            resident = list(HostImageFeeder(raw, pconf)) if tag.startswith("resident") else None
            for rep in range(3):  # first pass untimed
                feeder = HostImageFeeder(raw, pconf, view_key=raw_key if dedupe else None) if resident is None else None
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                out = []
                ep._export_loop(enumerate(feeder if resident is None else resident), pipe, "cuda", keys, optional, None, False, 1,
                                out, 32, item_key if dedupe else None)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
                if rep:
                    best = dt if best is None else min(best, dt)
            host[tag] = {"pairs_per_s": round(n_host / best, 1),
                         "matches_total": sum(int((rec["matches0"] >= 0).sum()) for _, _, rec in out)}
            if feeder is not None:
                host[tag]["h2d_mb_per_pair"] = round(feeder.h2d_bytes / n_host / 1e6, 2)
        host["same_integers"] = len({v["matches_total"] for v in host.values()}) == 1
        host["sample"] = (f"{n_host} pairs ({n_host // 4} distinct, walked four times; sequences of five share their view 0 as in the "
                          f"HPatches list) of decoded RGB uint8 images at original sizes {synthetic.HPATCHES_LIKE_ORIGINALS} (pinned "
                          "host memory), ImagePreprocessor resize 480 / side short on the GPU (HostImageFeeder), then the pair_batch32 "
                          "export loop; `view_dedupe` = the shared image is copied, resized and extracted once per occurrence window "
                          "(view_key on the feeder and on the loop); `resident_*` = the same preprocessed images already in HBM")
        res["from_host_uint8"] = host
        try:  # the same list FROM FILES: directory of binary PPMs -> eval_hpatches (read into pinned memory, resize on
            # the GPU, pair batches, predictions.h5 written) -> evaluation pass (match metrics + DLT on the GPU)
            res["from_files"] = c3_from_files(pipe)
        except Exception as e:  # noqa: BLE001 -- information only
            res["from_files"] = {"pairs_per_s": None, "error": repr(e)[:200]}
    res["same_match_count"] = len({v["matches_total"] for v in legs.values()}) == 1
    res["all_legs_integers_equal"] = all(v["integers_equal_to_first_leg"] for v in legs.values())
    res["legs"] = legs
    matches = legs["sequential"]["matches_total"]
    shapes = sorted({tuple(it[v]["image"].shape[-2:]) for it in items for v in ("view0", "view1")})
    return {"unit": "image-pairs/sec", **res, "matches_total": matches,
            "sample": f"{n_pairs} HPatches-shaped RGB pairs, image shapes {shapes} in changing combinations, official "
                      "SuperPoint + LightGlue (superpoint+lightglue-official configuration), 1024 key points; the whole "
                      "export loop per pair except the file write; best of two passes after one untimed pass; "
                      "`sequential_reference_syncs` keeps the reference's per-call device synchronisations "
                      "(two_view_pipeline.py:78-102), the other legs run with profile_calls: false"}


def c3_from_files(pipe, n_pairs: int = 240):
    """Information only: BASELINE config 3 from a directory of image files (datasets/hpatches.py:94-112 +
    eval/hpatches.py:98-176): `n_pairs` pairs in sequences of five written as binary PPMs at HPatches-like sizes to a
    temporary directory, then glue_factory_colon_amd.eval_hpatches end to end -- files read into pinned memory by reader
    threads, resize on the GPU, pair batches of 32 with the shared reference image handled once, `predictions.h5`
    written, then match metrics and DLT error for every pair on the GPU."""
    import shutil
    import tempfile

    from glue_factory_colon_amd import eval_hpatches

    root = tempfile.mkdtemp(prefix="gfc_bench_hp_")
    try:
        raw = synthetic.hpatches_like_host_images(n_pairs, seed=7000, pin=False, shared_view0=True)
        nbytes = 0
        for i, it in enumerate(raw):
            d = os.path.join(root, "hpatches-sequences-release", "v_" + it["scene"])
            os.makedirs(d, exist_ok=True)
            for name, img in ((("1.ppm", it["view0"]["image"]),) if i % 5 == 0 else ()) + ((f"{i % 5 + 2}.ppm", it["view1"]["image"]),):
                a = img.numpy()
                with open(os.path.join(d, name), "wb") as f:
                    f.write(b"P6\n" + f"{a.shape[1]} {a.shape[0]}\n255\n".encode() + a.tobytes())
                nbytes += a.size
            with open(os.path.join(d, f"H_1_{i % 5 + 2}"), "w") as f:
                f.write("1 0 0\n0 1 0\n0 0 1\n")
        del raw
        hp = eval_hpatches.HPatchesPipeline({"data_dir": os.path.join(root, "hpatches-sequences-release")}, pair_batch=32,
                                            num_workers=2)
        best = None
        for rep in range(3):  # first pass untimed (file cache, pinned-memory pool)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred = hp.get_predictions(os.path.join(root, "exp"), pipe, overwrite=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep:
                best = dt if best is None else min(best, dt)
        t0 = time.perf_counter()
        summaries, results = hp.run_eval(pred)
        t_eval = time.perf_counter() - t0
        return {"pairs_per_s": round(n_pairs / best, 1), "evaluation_pairs_per_s": round(n_pairs / t_eval, 1),
                "matches_total": int(sum(results["num_matches"])), "mb_of_files": round(nbytes / 1e6),
                "sample": f"{n_pairs} pairs in {n_pairs // 5} sequences as binary PPM files (page cache), "
                          "eval_hpatches.HPatchesPipeline.get_predictions incl. the predictions.h5 write, best of two "
                          "passes after one untimed pass; then run_eval (metrics + DLT on the GPU, grouped)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def c4_shape(dev, pairs: int = 32, steps: int = 8, warmup: int = 2):
    """Information only (never `value`): BASELINE configs[3]'s per-GPU shape -- 32 pairs of 1024 x 1024 images, 2048 key
    points -- the same step as the headline workload, with the attention launches timed by HIP events."""
    h = w = 1024
    k = 2048
    ext = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": k, "detection_threshold": 0.0,
                                      "nms_radius": 3, "force_num_keypoints": True}).eval().to(dev)
    mat = lightglue.LightGlue({"weights": "synthetic", "filter_threshold": 0.1, "depth_confidence": -1,
                               "width_confidence": -1}).eval().to(dev)
    v0, v1 = synthetic.synthetic_pairs(pairs, h, w, seed=1234, device=dev)
    size = torch.tensor([[float(w), float(h)]] * pairs, device=dev)
    both = {"image": torch.cat([v0, v1], 0), "image_size": torch.cat([size, size], 0)}
    del v0, v1

    def step():
        pj = ext(both)
        return mat({"keypoints0": pj["keypoints"][:pairs], "keypoints1": pj["keypoints"][pairs:],
                    "descriptors0": pj["descriptors"][:pairs], "descriptors1": pj["descriptors"][pairs:],
                    "view0": {"image_size": size}, "view1": {"image_size": size}})

    with torch.no_grad():
        for _ in range(warmup):
            step()
        atrace = nat.KernelTrace(2 * mat.conf.n_layers * steps)
        mat.trace = atrace
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            pred = step()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        mat.trace = None
    adurs = atrace.durations_ms()
    atrace.close()
    prod = 2.0 * k * k * 64 * 4
    n_self, n_cross = len(adurs[0::2]), len(adurs[1::2])
    alg = n_self * 2 * pairs * 2 * prod + n_cross * pairs * 3 * prod
    ach = alg / (sum(adurs) * 1e-3) / 1e12 if adurs else 0.0
    return {"pairs_per_s": round(pairs * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 2), "pairs_per_step": pairs,
            "image": [h, w], "keypoints": k, "steps": steps, "mean_matches_per_pair": round(float((pred["matches0"] >= 0).sum()) / pairs, 1),
            "attention": {"achieved_tflops": round(ach, 2), "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                          "avg_launch_ms": round(sum(adurs) / max(len(adurs), 1), 4), "launches_timed": len(adurs),
                          "share_of_step": round(sum(adurs) * 1e-3 / dt, 4),
                          "definition": "as `roofline`: algorithmic attention FLOPs (one shared cross sim) / HIP-event time / 157.3"},
            "pipeline_algorithmic_tflops": round(pairs * steps / dt * 580.6e9 / 1e12, 2)}


def config5(dev, batch: int = 4, pairs: int = 8, iters: int = 6):
    """Information only (never `value`): BASELINE configs[4] -- the DISK extractor (kornia's thin U-Net restated on HIP:
    parity unpinned, kornia absent) at VGA, and DISK + LightGlue with 128-d descriptors."""
    from glue_factory_colon_amd import disk_kornia, lightglue_pretrained

    k = 2048
    ext = disk_kornia.DISK({"weights": "synthetic", "max_num_keypoints": k, "force_num_keypoints": True,
                            "chunk": batch}).eval().to(dev)
    mat = lightglue_pretrained.LightGlue({"features": "disk", "weights": "synthetic", "filter_threshold": 0.1}).eval().to(dev)
    # (16, 16): DISK's U-Net has stride 16, so only such displacements keep its features equivariant (the (16, 8) shift of
    # the SuperPoint workloads left this leg with 143 matches of 2048 key points in round 4)
    g0, g1 = synthetic.synthetic_pairs(pairs, H, W, seed=77, dx=16, dy=16, device=dev)
    rgb0 = torch.cat([g0 * 0.8, g0, g0 * 0.9], 1).contiguous()
    rgb1 = torch.cat([g1 * 0.8, g1, g1 * 0.9], 1).contiguous()
    size = torch.tensor([[float(W), float(H)]] * pairs, device=dev)

    def timed(fn, n):
        for _ in range(2):
            out = fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n, out

    def pair_step():
        p0, p1 = ext({"image": rgb0}), ext({"image": rgb1})
        return mat({"keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"], "descriptors0": p0["descriptors"],
                    "descriptors1": p1["descriptors"], "view0": {"image_size": size}, "view1": {"image_size": size}})

    with torch.no_grad():
        t_net, _ = timed(lambda: ext.model.dense_nhwc(rgb0[:batch]), iters)
        t_ext, _ = timed(lambda: ext({"image": rgb0[:batch]}), iters)
        t_pair, pred = timed(pair_step, max(iters // 2, 2))
    layers = [(3, 16, 0), (16, 32, 1), (32, 64, 2), (64, 64, 3), (64, 64, 4), (128, 64, 3), (128, 64, 2), (96, 64, 1), (80, 129, 0)]
    flops = sum(2 * 25 * ci * co * (H >> lv) * (W >> lv) for ci, co, lv in layers)
    return {"disk_unet_ms_per_image": round(t_net * 1e3 / batch, 3),
            "disk_unet_algorithmic_tflops": round(flops * batch / t_net / 1e12, 1),
            "disk_extractor_ms_per_image": round(t_ext * 1e3 / batch, 3),
            "disk_lightglue128_pairs_per_s": round(pairs / t_pair, 1),
            "mean_matches_per_pair": round(float((pred["matches0"] >= 0).sum()) / pairs, 1),
            "sample": f"VGA RGB, {k} key points, extractor in chunks of {batch} images (disk_kornia.py:55-137); pairs/s = "
                      f"{pairs} pairs per step: DISK on both views + LightGlue with input_dim 128; name-seeded weights; "
                      "parity of the network unpinned (kornia absent offline: HIP vs the CPU restatement of its published source)"}


PER_RANK_FIELDS = ("busy_s", "elapsed_s", "attention_avg_launch_ms", "stem_avg_launch_ms", "probe_tflops",
                   "probe_shader_clock_ghz", "matches_last_step")


def per_rank_table(mine, world, device=None):
    """One all_gather AFTER the timed region: every rank's own numbers, so that an N > 1 line explains itself (a
    straggling rank, one GPU clocking lower with eight fp32-MFMA loads on the node, host contention).  `mine` follows
    PER_RANK_FIELDS.  busy_s = this rank's timed steps up to its own device synchronisation, BEFORE the closing barrier
    (`elapsed_s` includes the wait for the slowest rank; the job's `value` uses the MAX of it over ranks).
    Returns (list of per-rank dicts, {field: [min, median, max]})."""
    t = torch.tensor([float("nan") if v is None else float(v) for v in mine], dtype=torch.float64, device=device)
    if world > 1:
        rows = [torch.empty_like(t) for _ in range(world)]
        torch.distributed.all_gather(rows, t)
    else:
        rows = [t]
    table = [{"rank": r, **{k: (None if v != v else round(v, 5)) for k, v in zip(PER_RANK_FIELDS, row.tolist())}}
             for r, row in enumerate(rows)]
    summary = {}
    for k in PER_RANK_FIELDS:
        vals = sorted(row[k] for row in table if row[k] is not None)
        if vals:
            summary[k] = [vals[0], vals[len(vals) // 2] if len(vals) % 2 else round((vals[len(vals) // 2 - 1] + vals[len(vals) // 2]) / 2, 5),
                          vals[-1]]
    return table, summary


def host_barrier_group(world):
    """A gloo group for the job's closing barrier (rank 0 spends about a minute on the CPU baseline after the timed region:
    the other ranks wait on the host instead of spinning in an RCCL kernel).  One node only (the bench contract), so the
    loopback interface serves; if gloo cannot be set up the default (RCCL) group closes the job as before."""
    if world <= 1:
        return None
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    try:
        return torch.distributed.new_group(backend="gloo")
    except Exception as exc:  # noqa: BLE001  (every rank sees the same environment, hence the same outcome)
        print(f"bench: gloo group unavailable ({exc}); closing barrier on the default group", file=sys.stderr)
        return None

def conv_mode_of(arg):
    """The convolution arithmetic a module built with conf.conv_arithmetic = arg ends up with."""
    return arg if arg is not None else os.environ.get("GFC_CONV_MODE", "winograd")


def stem_is_f43(arg):
    """The default stem: conv1b as Winograd F(4x4,3x3), conv1a on the matrix pipe (csrc/conv_wino43.hip)."""
    return conv_mode_of(arg) == "winograd" and os.environ.get("GFC_STEM_F43", "1") != "0"


def stem_kernel_name(arg):
    mode = conv_mode_of(arg)
    if stem_is_f43(arg):
        return "stem_wino43_kernel (stem: conv1a on the matrix pipe + conv1b Winograd F(4x4,3x3) + ReLU + BN + 2x2 max-pool)"
    if mode == "winograd":
        return "conv3x3_wino_kernel<true, true> (stem: conv1a direct + conv1b Winograd F(2x2,3x3) + ReLU + BN + 2x2 max-pool)"
    return "conv3x3_mfma_kernel<true, true, 16, false> (stem: conv1a + conv1b 3x3 + ReLU + BN + 2x2 max-pool)"


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N rank processes (one per GPU) as children with the
    torchrun environment, wait for them and relay rank 0's stdout (the one JSON line).  The parent makes no GPU
    call (nothing before this point touches HIP), it only waits.  Returns the exit code for the parent."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    # a rank that dies (or never reaches the rendezvous) must not leave the parent waiting for ever: poll all ranks,
    # stop the others once one has failed, and give the whole job a deadline
    deadline = time.monotonic() + float(os.environ.get("GFC_BENCH_TIMEOUT_S", "1500"))
    import threading
    out0_chunks = []
    reader = threading.Thread(target=lambda: out0_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        failed = any(p.poll() not in (None, 0) for p in procs)
        if failed or time.monotonic() > deadline:
            time.sleep(2.0 if failed else 0.0)  # let the other ranks notice and exit by themselves first
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = "".join(c for c in out0_chunks if c)
    codes = [p.wait() for p in procs]
    for line in (out0 or "").splitlines():
        # rank 0's stdout carries the ONE JSON line; anything a backend library printed there goes to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        return bad[0][1] if bad[0][1] > 0 else 1
    return 0


def self_check(v0, v1, p0, p1, pred, idxs=(0, 10, 21, 31)):
    """Pairs `idxs` (spread over the batch) of the LAST timed step against the CPU oracle (outside the timed region):
    what was measured is also what is correct.  Same comparison as tests/test_gpu_batch32.py (which checks all 32)."""
    from oracle import lightglue as olg
    from oracle import superpoint as osp

    sd_sp, sd_lg = weights.superpoint_open_state_dict(0), weights.lightglue_state_dict(0)
    size = torch.tensor([[float(W), float(H)]])

    def pairs(kp0, kp1, m0, s0):
        kp0, kp1, m0, s0 = kp0.cpu(), kp1.cpu(), m0.cpu(), s0.cpu()
        return {(*kp0[a].tolist(), *kp1[int(m0[a])].tolist()): float(s0[a])
                for a in (m0 >= 0).nonzero().flatten().tolist()}

    idxs = [i for i in idxs if i < v0.shape[0]]
    tot = {"pairs": idxs, "keypoint_sets_equal": True, "pairs_equal": True, "matches": 0, "oracle_matches": 0,
           "pairs_common": 0, "score_err": 0.0, "checker": "oracle/ (PyTorch-CPU restatement of the reference path)"}
    for idx in idxs:
        imgs = torch.cat([v0[idx:idx + 1], v1[idx:idx + 1]], 0).cpu()
        o = osp.extract(sd_sp, imgs, "open", nms_radius=3, max_num_keypoints=K, detection_threshold=0.0)
        okp, ode = torch.stack(o["keypoints"]), torch.stack(o["descriptors"])
        ref = olg.match(sd_lg, okp[:1], okp[1:], ode[:1], ode[1:], size, size, filter_threshold=0.1)
        mine = pairs(p0["keypoints"][idx], p1["keypoints"][idx], pred["matches0"][idx], pred["matching_scores0"][idx])
        theirs = pairs(okp[0], okp[1], ref["matches0"][0], ref["matching_scores0"][0])
        common = set(mine) & set(theirs)
        kp_same = all(set(map(tuple, p["keypoints"][idx].cpu().tolist())) == set(map(tuple, okp[i].tolist()))
                      for i, p in enumerate((p0, p1)))
        tot["keypoint_sets_equal"] = bool(tot["keypoint_sets_equal"] and kp_same)
        tot["pairs_equal"] = bool(tot["pairs_equal"] and set(mine) == set(theirs))
        tot["matches"] += len(mine)
        tot["oracle_matches"] += len(theirs)
        tot["pairs_common"] += len(common)
        tot["score_err"] = max(tot["score_err"], max((abs(mine[q] - theirs[q]) for q in common), default=0.0))
    return tot


def rehearse_cpu(args):
    """Same control flow as main() around a dummy step, on gloo / CPU tensors (tests/test_host_cpu.py)."""
    rank, world, _ = sharding.init_from_env("gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    host_group = host_barrier_group(world)  # as main(): the closing barrier
    b = args.pairs
    pred = {"matches0": torch.full((b, K), -1, dtype=torch.long), "matching_scores0": torch.zeros(b, K),
            "keypoints0": torch.zeros(b, K, 2), "keypoints1": torch.zeros(b, K, 2)}
    pred["matches0"][:, : 10 + rank] = 1
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))  # rank r is r times slower: the per-rank table must show it
    busy = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
    elapsed_local = elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    gathered = sharding.gather_records(sharding.pack_pair_records(pred, K))
    per_rank, per_rank_summary = per_rank_table([busy, elapsed_local, 0.5 + 0.01 * rank, 5.0, None, None, 10 + rank], world)
    if rank == 0:
        allrec = torch.cat(gathered)
        # like main(): rank 0 times the CPU baseline after the gather while the other ranks wait at the last barrier
        base = None if args.no_cpu_baseline else cpu_baseline(args.cpu_pairs, args.cpu_iters)
        print(json.dumps({"metric": METRIC, "value": None, "unit": "image-pairs/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "rehearsal": True,
                          "pairs_gathered": int(allrec.shape[0]),
                          "matches_per_rank": [int(v) for v in allrec[::b, 0].tolist()],
                          "config": {"per_rank": per_rank, "per_rank_summary": per_rank_summary, "rccl_ranks": 0},
                          "cpu_baseline": base}), flush=True)
    if world > 1:
        torch.distributed.barrier(group=host_group)
        torch.distributed.destroy_process_group()


def main():
    global H, W, K, STEM_FLOPS_PER_IMAGE, PAIR_FLOPS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=32, help="image pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-self-check", action="store_true",
                    help="skip the oracle check of four pairs of the last timed step (after the timed region)")
    ap.add_argument("--cpu-pairs", type=int, default=2)
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--workload", default="c2", choices=["c2", "c4"],
                    help="c2 = BASELINE.json configs[1] (640x480, 1024 kpts; the headline metric); "
                         "c4 = configs[3] shape (1024x1024, 2048 kpts) for information")
    ap.add_argument("--joint-extract", type=int, default=1,
                    help="1: run the extractor once on both views' images (2*pairs images per call)")
    ap.add_argument("--no-experimental", action="store_true",
                    help="accepted for older scripts (the split-arithmetic leg was retired in round 4)")
    ap.add_argument("--no-batch1", action="store_true",
                    help="skip the informational legs (`batch1` single-pair latency, `c3_regime`, `c4`, `config5`)")
    ap.add_argument("--conv-arithmetic", default=None, choices=[None, "fp32", "winograd"],
                    help="3x3 convolutions of the timed path: Winograd F(2x2,3x3) / F(4x4,3x3) stem on fp32 MFMA (default) "
                         "or the direct implicit GEMM on fp32 MFMA")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU work: exercise only the multi-process plumbing (rendezvous, barriers, max-reduce, "
                         "final gather, JSON) on gloo with a dummy step; never a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process becomes the parent of the N ranks (it has made no GPU call and makes none)
        raise SystemExit(spawn_ranks(args.gpus))
    if args.rehearse_cpu:
        return rehearse_cpu(args)
    rank, world, local = sharding.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    nat.lib()  # fail loudly if the HIP library is missing
    # closing synchronisation on the HOST (gloo): rank 0 spends ~a minute after the timed region on its CPU baseline and
    # checks; behind an RCCL barrier the other ranks' GPUs would spin in a collective kernel for that long
    host_group = host_barrier_group(world)

    ext = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": K, "detection_threshold": 0.0,
                                      "nms_radius": 3, "force_num_keypoints": True,
                                      "conv_arithmetic": args.conv_arithmetic}).eval().to(dev)
    mat = lightglue.LightGlue({"weights": "synthetic", "filter_threshold": 0.1, "depth_confidence": -1,
                               "width_confidence": -1}).eval().to(dev)
    if args.workload == "c4":
        H, W, K = 1024, 1024, 2048
        STEM_FLOPS_PER_IMAGE = 2 * 9 * (1 * 64 + 64 * 64) * H * W
        PAIR_FLOPS = 580.6e9
        ext = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": K, "detection_threshold": 0.0,
                                          "nms_radius": 3, "force_num_keypoints": True,
                                          "conv_arithmetic": args.conv_arithmetic}).eval().to(dev)
    b = args.pairs
    v0, v1 = synthetic.synthetic_pairs(b, H, W, seed=1234 + rank, device=dev)  # resident in HBM
    size = torch.tensor([[float(W), float(H)]] * b, device=dev)
    view0, view1 = {"image": v0, "image_size": size}, {"image": v1, "image_size": size}

    both = {"image": torch.cat([v0, v1], 0), "image_size": torch.cat([size, size], 0)}

    def step(ext=ext, mat=mat):
        if args.joint_extract:
            # both views through ONE extractor call (images are independent: identical results, fewer launches)
            pj = ext(both)
            p0 = {k: v[:b] for k, v in pj.items()}
            p1 = {k: v[b:] for k, v in pj.items()}
        else:
            p0, p1 = ext(view0), ext(view1)
        return p0, p1, mat({"keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"],
                            "descriptors0": p0["descriptors"], "descriptors1": p1["descriptors"],
                            "view0": view0, "view1": view1})

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    calls = 1 if args.joint_extract else 2
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        trace = nat.KernelTrace(calls * args.steps)                     # stem launches
        atrace = nat.KernelTrace(2 * mat.conf.n_layers * args.steps)   # attention launches (self, cross per layer)
        ext._runner.trace = trace
        mat.trace = atrace
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            p0, p1, pred = step()
        torch.cuda.synchronize(dev)
        busy = time.perf_counter() - t0  # this rank's own steps; the closing barrier waits for the slowest rank
        barrier()
        elapsed_local = elapsed = time.perf_counter() - t0
        ext._runner.trace = None
        mat.trace = None

    # max over ranks
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    # the one collective of the path: gather of fixed-size per-pair records to rank 0
    rec = sharding.pack_pair_records({**pred, "keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"]}, K)
    tg = time.perf_counter()
    gathered = sharding.gather_records(rec)
    torch.cuda.synchronize(dev)
    gather_ms = (time.perf_counter() - tg) * 1e3

    durs = trace.durations_ms()
    adurs = atrace.durations_ms()
    trace.close()
    atrace.close()

    # what the matrix pipe of THIS rank's GPU sustains right after the timed region (the data-sheet 157.3 TFLOP/s assumes
    # 2.4 GHz), and every rank's own timings: one all_gather, outside the timed region
    try:
        tf, ghz = ctypes.c_float(0), ctypes.c_float(0)
        nat.check(nat.lib().gfc_probe_mfma_peak(80000, ctypes.byref(tf), ctypes.byref(ghz), nat.stream_ptr(dev)), "probe")
        probe = {"tflops": round(tf.value, 1), "shader_clock_ghz": round(ghz.value, 3),
                 "note": "registers-only v_mfma_f32_32x32x2_f32 loop on every SIMD, measured after the timed region"}
    except Exception as e:  # noqa: BLE001
        probe = {"tflops": None, "shader_clock_ghz": None, "error": repr(e)[:120]}
    per_rank, per_rank_summary = per_rank_table(
        [busy, elapsed_local, sum(adurs) / max(len(adurs), 1), sum(durs) / max(len(durs), 1), probe["tflops"],
         probe["shader_clock_ghz"], int((pred["matches0"] >= 0).sum())], world, dev)

    if rank == 0:
        allrec = torch.cat(gathered)
        n_pairs_total = allrec.shape[0]
        mean_matches = float(allrec[:, 0].mean())
        value = world * b * args.steps / elapsed
        wino = conv_mode_of(args.conv_arithmetic) == "winograd"
        default_shape = args.workload == "c2" and b == 32 and args.joint_extract and wino

        # ---- dominant kernel: attention (SURVEY.md 8d "attention-MFMA roofline") ------------------------------
        # one product = Q.K^T or P.V of one problem over its 4 heads: 2 * K * K * 64 * 4 FLOP
        prod = 2.0 * K * K * 64 * 4
        self_alg = 2 * b * 2 * prod          # 2b images, QK^T + PV each
        cross_alg = b * 3 * prod             # b pairs: ONE sim + two PV products (lightglue.py:207-217)
        cross_exec = b * 4 * prod            # the kernel evaluates the second direction's sim again
        self_ms, cross_ms = adurs[0::2], adurs[1::2]
        n_self, n_cross = len(self_ms), len(cross_ms)
        att_ms = sum(adurs)
        att_alg = n_self * self_alg + n_cross * cross_alg
        att_exec = n_self * self_alg + n_cross * cross_exec

        def tflops(flops, ms):
            return flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

        att_kernel = ("attention_kernel<2, 4>" if 2 * b * 4 * ((K + 255) // 256) >= 1024 else "attention_kernel<1, 4>")
        att_traffic, att_src = pmc_traffic_bytes("attention_kernel<2, 4>") if default_shape else (None, None)
        ach = tflops(att_alg, att_ms)
        roof = {"bound": "mfma",
                "kernel": att_kernel + " (flash-style self / bidirectional cross attention, fp32 MFMA 32x32x2; the "
                          "kernel with the largest share of the step)",
                "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                "executed_tflops": round(tflops(att_exec, att_ms), 2),
                "executed_frac": round(tflops(att_exec, att_ms) / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": att_traffic,
                "traffic_note": ("HBM bytes/launch (mean of self and cross launches), rocprofv3 --pmc FETCH_SIZE(x2) + "
                                 f"WRITE_SIZE passes of the default command ({att_src}); algorithmic per launch: Q, K, V "
                                 f"read once + O written = {2 * b * K * 256 * 4 * 4 / 1e6:.1f} MB"),
                "launches_timed": len(adurs),
                "avg_launch_ms": round(att_ms / max(len(adurs), 1), 4),
                "flops_per_launch": round(att_alg / max(len(adurs), 1)),
                "share_of_step": round(att_ms / (elapsed * 1e3), 4),
                "definition": "achieved = ALGORITHMIC attention FLOPs (QK^T + PV; cross attention counted with ONE "
                              "shared sim) summed over the timed launches / their summed HIP-event durations; "
                              "executed_* counts the cross sim twice, as the kernel evaluates it"}

        # ---- the other traced kernels ---------------------------------------------------------------------------
        kernels = [
            {"kernel": att_kernel + " [self launches]", "launches_timed": n_self,
             "avg_launch_ms": round(sum(self_ms) / max(n_self, 1), 4), "flops_per_launch": round(self_alg),
             "achieved": round(tflops(n_self * self_alg, sum(self_ms)), 2),
             "frac": round(tflops(n_self * self_alg, sum(self_ms)) / FP32_MFMA_PEAK_TFLOPS, 4)},
            {"kernel": att_kernel + " [cross launches]", "launches_timed": n_cross,
             "avg_launch_ms": round(sum(cross_ms) / max(n_cross, 1), 4), "flops_per_launch": round(cross_alg),
             "achieved": round(tflops(n_cross * cross_alg, sum(cross_ms)), 2),
             "frac": round(tflops(n_cross * cross_alg, sum(cross_ms)) / FP32_MFMA_PEAK_TFLOPS, 4),
             "executed_frac": round(tflops(n_cross * cross_exec, sum(cross_ms)) / FP32_MFMA_PEAK_TFLOPS, 4)},
        ]
        stem_ms = sum(durs)
        imgs_per_launch = 2 * b if args.joint_extract else b
        stem_alg = STEM_FLOPS_PER_IMAGE * imgs_per_launch  # direct-convolution FLOPs of SURVEY.md 8d
        # Winograd F(2x2,3x3): 16 instead of 36 multiplications per 2x2 output block and input channel, so the matrix
        # pipe executes 4/9 of conv1b's direct FLOPs (conv1a, cin = 1, runs beside it on the VALU).
        # F(4x4,3x3) stem (default): 36 per 4x4 block = 1/4 of conv1b's direct FLOPs, plus conv1a on the pipe: per
        # 32x16-pixel work item 36 positions x 2 cout tiles x 32 k steps + 40 conv1a units x 5 = 2504 MFMAs of 4096 FLOP
        if stem_is_f43(args.conv_arithmetic):
            stem_exec = 2504 * 4096 * ((H + 15) // 16) * ((W + 31) // 32) * imgs_per_launch
        else:
            stem_exec = (2 * 4 * 64 * 64 * H * W * imgs_per_launch) if wino else stem_alg
        stem_traffic, stem_src = (pmc_traffic_bytes("stem_wino43_kernel" if stem_is_f43(args.conv_arithmetic)
                                                    else "conv3x3_wino_kernel<true, true")
                                  if default_shape else (None, None))
        kernels.append(
            {"kernel": stem_kernel_name(args.conv_arithmetic), "launches_timed": len(durs),
             "avg_launch_ms": round(stem_ms / max(len(durs), 1), 4),
             "executed_flops_per_launch": stem_exec,
             "achieved": round(tflops(len(durs) * stem_exec, stem_ms), 2),
             "frac": round(tflops(len(durs) * stem_exec, stem_ms) / FP32_MFMA_PEAK_TFLOPS, 4),
             "algorithmic_equiv_tflops": round(tflops(len(durs) * stem_alg, stem_ms), 2),
             "share_of_step": round(stem_ms / (elapsed * 1e3), 4),
             "traffic": stem_traffic,
             "note": "frac = FLOPs the matrix pipe EXECUTES / peak; algorithmic_equiv_tflops = direct-convolution FLOPs "
                     "(SURVEY.md 8d) / time: a Winograd kernel may exceed the pipe's peak in that unit, it is not a "
                     f"roofline fraction; traffic from {stem_src}, algorithmic "
                     f"{imgs_per_launch * H * W * 4 / 1e6:.1f} MB in + "
                     f"{imgs_per_launch * (H // 2) * (W // 2) * 64 * 4 / 1e6:.1f} MB out"})
        roof["kernels"] = kernels

        exec_pair_flops = PAIR_FLOPS_EXECUTED if args.workload == "c2" and stem_is_f43(args.conv_arithmetic) else None
        out = {
            "metric": METRIC if args.workload == "c2" else "image-pairs/sec (SuperPoint+LightGlue, 2048 kpts, 1024x1024)",
            "value": round(value, 3),
            "unit": "image-pairs/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("SuperPoint-open + LightGlue, 1024 kpts, 640x480 synthetic pairs "
                                    "(BASELINE.json configs[1])") if args.workload == "c2" else
                                   ("SuperPoint-open + LightGlue, 2048 kpts, 1024x1024 synthetic pairs "
                                    "(BASELINE.json configs[3] shape; NOT the headline metric)"),
                       "pairs_per_gpu_per_step": b,
                       "global_pairs_per_step": b * world, "keypoints": K, "image": [H, W],
                       "parallelism": f"dp{world} (pairs sharded, one final gather)",
                       "weights": "name-seeded seed 0 (no network)", "mean_matches_per_pair": round(mean_matches, 1),
                       "pairs_gathered": n_pairs_total, "extractor_calls_per_step": calls,
                       "final_gather_ms": round(gather_ms, 3),
                       "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                       # every rank's own numbers (one all_gather after the timed region): `value` uses the MAX of
                       # elapsed_s; busy_s excludes the closing barrier's wait for the slowest rank
                       "per_rank": per_rank, "per_rank_summary": per_rank_summary,
                       # whole path per GPU: direct-arithmetic FLOPs of SURVEY.md 8d (a Winograd / folded implementation
                       # executes fewer), and the FLOPs the matrix pipe really executes
                       "pipeline_algorithmic_tflops": round(value / world * PAIR_FLOPS / 1e12, 2),
                       "pipeline_executed_tflops": (round(value / world * exec_pair_flops / 1e12, 2)
                                                    if exec_pair_flops else None),
                       "pipeline_executed_frac": (round(value / world * exec_pair_flops / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
                                                  if exec_pair_flops else None)},
            "roofline": roof,
        }
        out["roofline"]["sustained_mfma_probe"] = probe  # rank 0's; every rank's is in config.per_rank
        if args.workload == "c2" and not args.no_self_check:
            try:
                out["self_check"] = self_check(v0, v1, p0, p1, pred)
            except Exception as e:  # noqa: BLE001
                out["self_check"] = {"pairs_equal": None, "error": repr(e)[:200]}
        if args.workload == "c2" and not args.no_batch1:
            try:  # information only: the batch-1 regime of the HPatches evaluation loop (never `value`)
                out["batch1"] = batch1_latency(dev)
            except Exception as e:  # noqa: BLE001
                out["batch1"] = {"pairs_per_s": None, "error": repr(e)[:200]}
        if args.workload == "c2" and not args.no_batch1 and world == 1:
            try:  # information only: config 3's regime at the batched rate (never `value`)
                out["c3_regime"] = c3_regime(dev)
            except Exception as e:  # noqa: BLE001
                out["c3_regime"] = {"pair_batch32": None, "error": repr(e)[:200]}
        if args.workload == "c2" and not args.no_batch1 and world == 1:
            for key, leg in (("c4", c4_shape), ("config5", config5)):
                try:  # information only: the other BASELINE configs' shapes (never `value`)
                    out[key] = leg(dev)
                except Exception as e:  # noqa: BLE001
                    out[key] = {"pairs_per_s": None, "error": repr(e)[:200]}
                torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            # rank 0 only, after the timed region and the gather; at N > 1 the other ranks wait at the closing barrier
            out["cpu_baseline"] = cpu_baseline(args.cpu_pairs, args.cpu_iters)
            if args.workload == "c2" and world == 1:
                try:  # context only: never let the comparison leg cost the measurement line
                    out["cpu_baseline"]["same_gpu_torch_eager"] = torch_eager_same_gpu(dev)
                except Exception as e:  # noqa: BLE001
                    out["cpu_baseline"]["same_gpu_torch_eager"] = {"value": None, "error": repr(e)[:200]}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier(group=host_group)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

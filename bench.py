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

Rank 0 prints ONE JSON line with `roofline` (dominant kernel: the fp32-MFMA stem
convolution conv1a+conv1b+pool, timed live with HIP events around each launch inside the timed region) and
`cpu_baseline` (the CPU oracle, a PyTorch-CPU port of the reference path, on a bounded sample).
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
PAIR_FLOPS = 182.3e9  # whole path per pair at this configuration
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def pmc_traffic_bytes():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same
    command (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes; tools/pmc_summary.py).  PMC counters
    cannot be collected from inside the process, so the figure is read from profiles/; None if absent."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_summary.json")
    try:
        with open(path) as f:
            summary = json.load(f)
        key = next(k for k in summary if k.startswith("void conv3x3_wino_kernel<true, true")
                   or k.startswith("conv3x3_wino_kernel<true, true"))  # the stem variant of the default path
        return summary[key]["hbm_bytes_per_launch"]  # 64-image launches
    except (OSError, KeyError, ValueError, StopIteration):
        return None


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


def conv_mode_of(arg):
    """The convolution arithmetic a module built with conf.conv_arithmetic = arg ends up with."""
    return arg if arg is not None else os.environ.get("GFC_CONV_MODE", "winograd")


def stem_kernel_name(arg):
    mode = conv_mode_of(arg)
    if mode == "winograd":
        return "conv3x3_wino_kernel<true, true> (stem: conv1a direct + conv1b Winograd F(2x2,3x3) + ReLU + BN + 2x2 max-pool)"
    if mode == "split":
        return "conv3x3_split_kernel (stem, experimental split arithmetic)"
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
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for line in (out0 or "").splitlines():
        # rank 0's stdout carries the ONE JSON line; anything a backend library printed there goes to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: rank(s) failed: {bad}", file=sys.stderr)
        return bad[0][1] if bad[0][1] > 0 else 1
    return 0


def self_check(v0, v1, p0, p1, pred, idx=0):
    """Pair `idx` of the LAST timed step against the CPU oracle (outside the timed region): what was measured is
    also what is correct.  Same comparison as tests/test_gpu_batch32.py."""
    from oracle import lightglue as olg
    from oracle import superpoint as osp

    imgs = torch.cat([v0[idx:idx + 1], v1[idx:idx + 1]], 0).cpu()
    o = osp.extract(weights.superpoint_open_state_dict(0), imgs, "open", nms_radius=3, max_num_keypoints=K,
                    detection_threshold=0.0)
    okp, ode = torch.stack(o["keypoints"]), torch.stack(o["descriptors"])
    size = torch.tensor([[float(W), float(H)]])
    ref = olg.match(weights.lightglue_state_dict(0), okp[:1], okp[1:], ode[:1], ode[1:], size, size,
                    filter_threshold=0.1)

    def pairs(kp0, kp1, m0, s0):
        kp0, kp1, m0, s0 = kp0.cpu(), kp1.cpu(), m0.cpu(), s0.cpu()
        return {(*kp0[a].tolist(), *kp1[int(m0[a])].tolist()): float(s0[a])
                for a in (m0 >= 0).nonzero().flatten().tolist()}

    mine = pairs(p0["keypoints"][idx], p1["keypoints"][idx], pred["matches0"][idx], pred["matching_scores0"][idx])
    theirs = pairs(okp[0], okp[1], ref["matches0"][0], ref["matching_scores0"][0])
    common = set(mine) & set(theirs)
    kp_same = all(set(map(tuple, p["keypoints"][idx].cpu().tolist())) == set(map(tuple, okp[i].tolist()))
                  for i, p in enumerate((p0, p1)))
    return {"pair": idx, "keypoint_sets_equal": bool(kp_same), "pairs_equal": set(mine) == set(theirs),
            "matches": len(mine), "oracle_matches": len(theirs), "pairs_common": len(common),
            "score_err": max((abs(mine[q] - theirs[q]) for q in common), default=0.0),
            "checker": "oracle/ (PyTorch-CPU restatement of the reference path)"}


def rehearse_cpu(args):
    """Same control flow as main() around a dummy step, on gloo / CPU tensors (tests/test_host_cpu.py)."""
    rank, world, _ = sharding.init_from_env("gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    b = args.pairs
    pred = {"matches0": torch.full((b, K), -1, dtype=torch.long), "matching_scores0": torch.zeros(b, K),
            "keypoints0": torch.zeros(b, K, 2), "keypoints1": torch.zeros(b, K, 2)}
    pred["matches0"][:, : 10 + rank] = 1
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    gathered = sharding.gather_records(sharding.pack_pair_records(pred, K))
    if rank == 0:
        allrec = torch.cat(gathered)
        print(json.dumps({"metric": METRIC, "value": None, "unit": "image-pairs/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "rehearsal": True,
                          "pairs_gathered": int(allrec.shape[0]),
                          "matches_per_rank": [int(v) for v in allrec[::b, 0].tolist()]}), flush=True)
    if world > 1:
        torch.distributed.barrier()
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
                    help="skip the oracle check of pair 0 of the last timed step (after the timed region)")
    ap.add_argument("--cpu-pairs", type=int, default=2)
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--workload", default="c2", choices=["c2", "c4"],
                    help="c2 = BASELINE.json configs[1] (640x480, 1024 kpts; the headline metric); "
                         "c4 = configs[3] shape (1024x1024, 2048 kpts) for information")
    ap.add_argument("--joint-extract", type=int, default=1,
                    help="1: run the extractor once on both views' images (2*pairs images per call)")
    ap.add_argument("--no-experimental", action="store_true",
                    help="skip the extra `experimental_split_arithmetic` leg (profiling runs: only the default path's kernels)")
    ap.add_argument("--linear-arithmetic", default=None, choices=[None, "fp32", "split"],
                    help="LightGlue GEMMs of the timed path (see --conv-arithmetic)")
    ap.add_argument("--conv-arithmetic", default=None, choices=[None, "fp32", "split", "winograd"],
                    help="3x3 convolutions of the timed path: fp32 MFMA (default) or the experimental bf16x3-split MFMA "
                         "products at fp32 accuracy; the default run additionally reports the split variant as "
                         "`experimental_split_arithmetic` (N = 1 only)")
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

    ext = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": K, "detection_threshold": 0.0,
                                      "nms_radius": 3, "force_num_keypoints": True,
                                      "conv_arithmetic": args.conv_arithmetic}).eval().to(dev)
    mat = lightglue.LightGlue({"weights": "synthetic", "filter_threshold": 0.1, "depth_confidence": -1,
                               "width_confidence": -1, "linear_arithmetic": args.linear_arithmetic,
                               "attention_arithmetic": args.linear_arithmetic}).eval().to(dev)
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

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        trace = nat.KernelTrace(2 * args.steps)
        ext._runner.trace = trace
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            p0, p1, pred = step()
        torch.cuda.synchronize(dev)
        barrier()
        elapsed = time.perf_counter() - t0
        ext._runner.trace = None

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
    trace.close()

    # information only: the same steps with the experimental split-bf16 convolutions (opt-in arithmetic, not `value`)
    split_info = None
    if (world == 1 and args.conv_arithmetic is None and args.linear_arithmetic is None and args.workload == "c2"
            and not args.no_experimental):
        try:
            ext_s = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": K,
                                                "detection_threshold": 0.0, "nms_radius": 3, "force_num_keypoints": True,
                                                "conv_arithmetic": "split"}).eval().to(dev)
            mat_s = lightglue.LightGlue({"weights": "synthetic", "filter_threshold": 0.1, "depth_confidence": -1,
                                         "width_confidence": -1, "linear_arithmetic": "split",
                                         "attention_arithmetic": "split"}).eval().to(dev)
            with torch.no_grad():
                for _ in range(args.warmup):
                    step(ext_s, mat_s)
                torch.cuda.synchronize(dev)
                ts = time.perf_counter()
                for _ in range(args.steps):
                    _, _, pred_s = step(ext_s, mat_s)
                torch.cuda.synchronize(dev)
                dts = time.perf_counter() - ts
            same = (pred_s["matches0"] >= 0).sum().item(), (pred["matches0"] >= 0).sum().item()
            split_info = {"value": round(b * args.steps / dts, 3), "unit": "image-pairs/sec",
                          "ms_per_step": round(dts / args.steps * 1e3, 3), "matches_split_vs_fp32": list(same),
                          "note": "conv_arithmetic / linear_arithmetic / attention_arithmetic = 'split': 3x3 convolutions, "
                                  "the LightGlue GEMMs and the attention products as six bf16 MFMA products per fp32 "
                                  "product (three bf16 planes per operand, fp32 accumulate, fp32 soft-max).  fp32-level "
                                  "error, whole parity suite green (GFC_CONV_MODE=split GFC_LINEAR_MODE=split "
                                  "GFC_ATTN_MODE=split pytest -m gpu); opt-in, NOT the headline"}
        except Exception as e:  # noqa: BLE001
            split_info = {"value": None, "error": repr(e)[:200]}
    if rank == 0:
        allrec = torch.cat(gathered)
        n_pairs_total = allrec.shape[0]
        mean_matches = float(allrec[:, 0].mean())
        avg_ms = sum(durs) / max(len(durs), 1)
        imgs_per_launch = 2 * b if args.joint_extract else b
        flops_per_launch = STEM_FLOPS_PER_IMAGE * imgs_per_launch  # one stem launch per extractor call
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if durs else 0.0
        value = world * b * args.steps / elapsed
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
                       "pairs_gathered": n_pairs_total, "extractor_calls_per_step": 1 if args.joint_extract else 2, "final_gather_ms": round(gather_ms, 3),
                       "rccl_ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                       "pipeline_tflops": round(value / world * PAIR_FLOPS / 1e12, 2)},
            "roofline": {"bound": "mfma", "kernel": stem_kernel_name(args.conv_arithmetic),
                         "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                         # the committed PMC passes are of the default command; other shapes: not measured
                         "traffic": pmc_traffic_bytes() if (args.workload == "c2" and imgs_per_launch == 64
                                                            and conv_mode_of(args.conv_arithmetic) == "winograd") else None,
                         "traffic_note": "HBM bytes/launch, rocprofv3 --pmc FETCH_SIZE(x2)+WRITE_SIZE passes of the "
                                         "default command (profiles/r02_pmc_summary.json); algorithmic per launch: "
                                         f"{imgs_per_launch * H * W * 4 / 1e6:.1f} MB image in + "
                                         f"{imgs_per_launch * (H // 2) * (W // 2) * 64 * 4 / 1e6:.1f} MB pooled "
                                         "activation out",
                         "launches_timed": len(durs), "avg_launch_ms": round(avg_ms, 4),
                         "flops_per_launch": flops_per_launch},
        }
        if conv_mode_of(args.conv_arithmetic) == "winograd":
            # `achieved` above is ALGORITHMIC (direct-convolution FLOPs of SURVEY.md 8d / launch time), as the contract
            # defines it; the Winograd kernel issues 16 instead of 36 multiplications per 2x2 output block and cin, so
            # the matrix pipe executes 4/9 of conv1b's algorithmic FLOPs (conv1a runs on the VALU).
            executed = 2 * 4 * 64 * 64 * H * W * imgs_per_launch
            mfma_tf = executed / (avg_ms * 1e-3) / 1e12 if durs else 0.0
            out["roofline"]["mfma_executed_tflops"] = round(mfma_tf, 2)
            out["roofline"]["mfma_frac"] = round(mfma_tf / FP32_MFMA_PEAK_TFLOPS, 4)
            out["roofline"]["note"] = ("Winograd F(2x2,3x3) on fp32 MFMA: frac = algorithmic FLOPs / peak may exceed 1; "
                                       "mfma_frac = FLOPs the matrix pipe actually executes / peak")
        try:  # what the matrix pipe of THIS box sustains (the data-sheet 157.3 TFLOP/s assumes 2.4 GHz): context only
            tf, ghz = ctypes.c_float(0), ctypes.c_float(0)
            nat.check(nat.lib().gfc_probe_mfma_peak(80000, ctypes.byref(tf), ctypes.byref(ghz), nat.stream_ptr(dev)), "probe")
            out["roofline"]["sustained_mfma_probe"] = {
                "tflops": round(tf.value, 1), "shader_clock_ghz": round(ghz.value, 3),
                "note": "registers-only v_mfma_f32_32x32x2_f32 loop on every SIMD, measured after the timed region"}
        except Exception as e:  # noqa: BLE001
            out["roofline"]["sustained_mfma_probe"] = {"tflops": None, "error": repr(e)[:120]}
        if split_info is not None:
            out["experimental_split_arithmetic"] = split_info
        if args.conv_arithmetic == "split" or args.linear_arithmetic == "split":
            out["dtype"] = "f32 via 3 x bf16 split MFMA in the 3x3 convolutions (experimental), f32 elsewhere"
            out["roofline"]["note"] = "split arithmetic: the stem is not an fp32-MFMA kernel; frac is fp32-equivalent FLOPs / fp32 peak"
        if args.workload == "c2" and not args.no_self_check:
            try:
                out["self_check"] = self_check(v0, v1, p0, p1, pred)
            except Exception as e:  # noqa: BLE001
                out["self_check"] = {"pairs_equal": None, "error": repr(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_pairs, args.cpu_iters)
            if args.workload == "c2":
                try:  # context only: never let the comparison leg cost the measurement line
                    out["cpu_baseline"]["same_gpu_torch_eager"] = torch_eager_same_gpu(dev)
                except Exception as e:  # noqa: BLE001
                    out["cpu_baseline"]["same_gpu_torch_eager"] = {"value": None, "error": repr(e)[:200]}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

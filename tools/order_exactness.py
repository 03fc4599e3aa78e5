"""How far is each fp32 evaluation of the SuperPoint heat-map from the EXACT top-1024 order?

The top-k list is sorted by score (superpoint_open.py:54-58), and scores that the exact arithmetic separates by less than
fp32 round-off come out in an order that depends on the accumulation order of the convolutions.  `tests/golden/
c2_fp64_order.npz` holds the order in float64 arithmetic (the reference's module cast to double, made by
tests/golden/make_golden.py) for 16 images of the benchmarked C2 batch.  This tool counts, against that order:

  reference   torch-CPU fp32 = what the reference itself produces (tests/golden/c2_batch32.npz)
  hip_fp32    HIP, conv_arithmetic "fp32"   (direct fp32-MFMA convolutions everywhere)
  hip_default HIP, default                  (Winograd F(2x2,3x3) layers, F(4x4,3x3) stem layer 2)
  hip_f23stem HIP, GFC_STEM_F43=0           (Winograd F(2x2,3x3) everywhere)

Per variant: set differences against the exact top-1024 (flips at the selection boundary), points that sit at another
rank than in the exact order (displaced), pair inversions, the largest exact-score gap that any displaced point jumped
over, and the heat-map error at the selected points.  The library reads its knobs once per process, so every HIP variant
runs in a child process (started before this process touches the GPU; it never does).

    python tools/order_exactness.py [--out profiles/r06_order_exactness.json]
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
K = 1024

VARIANTS = {  # name -> (conv_arithmetic, extra environment)
    "hip_fp32": ("fp32", {}),
    "hip_default": (None, {}),
    "hip_f23stem": (None, {"GFC_STEM_F43": "0"}),
}


def child(conv_arithmetic, out_path):
    import torch

    from bench_inputs import FP64_IMAGES
    from glue_factory_colon_amd import superpoint_open, synthetic

    n = FP64_IMAGES // 2
    v0, v1 = synthetic.synthetic_pairs(32, 480, 640, seed=1234)
    imgs = torch.stack([v0[:n], v1[:n]], 1).reshape(2 * n, 1, 480, 640).cuda()  # pair-major: (pair, view)
    conf = {"weights": "synthetic", "max_num_keypoints": K, "detection_threshold": 0.0, "nms_radius": 3,
            "force_num_keypoints": True}
    if conv_arithmetic:
        conf["conv_arithmetic"] = conv_arithmetic
    ext = superpoint_open.SuperPoint(conf).eval().cuda()
    with torch.no_grad():
        p = ext({"image": imgs})
    torch.cuda.synchronize()
    np.savez(out_path, keypoints=(p["keypoints"] - 0.5).round().short().cpu().numpy().reshape(n, 2, K, 2),
             scores=p["keypoint_scores"].cpu().numpy().reshape(n, 2, K))


def order_stats(kp, sc, exact_kp, exact_sc):
    """kp [K,2] int, sc [K] of one variant; exact_kp [D,2], exact_sc [D] float64 (D >= K) in exact order."""
    rank = {(int(x), int(y)): r for r, (x, y) in enumerate(exact_kp.tolist())}
    exact_rank = np.array([rank.get((int(x), int(y)), -1) for x, y in kp.tolist()])
    flips = int(np.sum((exact_rank < 0) | (exact_rank >= K)))  # selected here but not in the exact top-K
    inside = exact_rank >= 0
    r = exact_rank[inside]
    # displaced: rank among the common points differs from the exact rank among the common points
    order = np.argsort(np.argsort(r, kind="stable"), kind="stable")  # exact rank among common, in my list order
    displaced = int(np.sum(order != np.arange(len(r))))
    inversions = int(sum(np.sum(r[i + 1:] < r[i]) for i in range(len(r))))
    gap = 0.0
    if displaced:
        s_exact = exact_sc[r]  # exact scores in my order; a sorted list would be non-increasing
        moved = np.nonzero(order != np.arange(len(r)))[0]
        by_exact = np.sort(s_exact)[::-1]
        gap = float(np.max(np.abs(s_exact[moved] - by_exact[moved])))  # exact score of the point vs of the rank it took
    err = float(np.max(np.abs(sc[inside].astype(np.float64) - exact_sc[r])))
    return {"flips": flips, "displaced": displaced, "inversions": inversions, "max_gap_jumped": gap, "score_err_vs_fp64": err}


def main():
    out_path = os.path.join(ROOT, "profiles", "r06_order_exactness.json")
    if "--out" in sys.argv:
        out_path = sys.argv[sys.argv.index("--out") + 1]
    gold = os.path.join(ROOT, "tests", "golden")
    exact = np.load(os.path.join(gold, "c2_fp64_order.npz"))
    ref = np.load(os.path.join(gold, "c2_batch32.npz"))
    n = exact["keypoints"].shape[0]
    lists = {"reference": (ref["keypoints"][:n], ref["keypoint_scores"][:n])}
    tmp = os.path.join(ROOT, "gpurun_out", "order_exactness")
    os.makedirs(tmp, exist_ok=True)
    for name, (arith, env) in VARIANTS.items():
        path = os.path.join(tmp, name + ".npz")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", arith or "default", path],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        if r.returncode:
            raise SystemExit(f"{name}: child failed\n{r.stderr[-2000:]}")
        z = np.load(path)
        lists[name] = (z["keypoints"], z["scores"])
    gaps = -np.diff(exact["scores"][:, :, :K], axis=-1)
    result = {"images": int(2 * n), "k": K,
              "exact_order_gaps": {"min": float(gaps.min()), "below_1e-6": int((gaps < 1e-6).sum()),
                                   "below_5e-6": int((gaps < 5e-6).sum()), "pairs_of_neighbours": int(gaps.size)},
              "variants": {}}
    for name, (kp, sc) in lists.items():
        per = [order_stats(kp[i, s], sc[i, s], exact["keypoints"][i, s], exact["scores"][i, s])
               for i in range(n) for s in (0, 1)]
        result["variants"][name] = {
            "flips_total": sum(p["flips"] for p in per),
            "displaced_total": sum(p["displaced"] for p in per),
            "displaced_per_image_min_max": [min(p["displaced"] for p in per), max(p["displaced"] for p in per)],
            "inversions_total": sum(p["inversions"] for p in per),
            "max_gap_jumped": max(p["max_gap_jumped"] for p in per),
            "score_err_vs_fp64": max(p["score_err_vs_fp64"] for p in per),
        }
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    if "--child" in sys.argv:
        i = sys.argv.index("--child")
        child(None if sys.argv[i + 1] == "default" else sys.argv[i + 1], sys.argv[i + 2])
    else:
        main()

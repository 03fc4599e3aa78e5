"""Randomised MODULE-level sweep on the GPU against the oracle (python tools/micro/fuzz_models.py [n_cases]): the two
SuperPoint variants on image sizes that are not multiples of 8, every NMS radius, border widths, thresholds, top-k above /
below the candidate count, RGB, batches (per-image calls and force_num_keypoints batches), official sampling modes with
image_size; LightGlue on tiny / unequal / batched key-point sets.  Key-point lists are compared with
tests/parity_utils.compare_keypoints (identical sets, every order swap a near tie), matches as index arrays."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from glue_factory_colon_amd import lightglue, superpoint, superpoint_open, synthetic, weights  # noqa: E402
from oracle import lightglue as olg  # noqa: E402
from oracle import superpoint as osp  # noqa: E402
from parity_utils import compare_keypoints  # noqa: E402


def run(n_cases=24, seed=99):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    bad = []
    sd_open, sd_off, sd_lg = weights.superpoint_open_state_dict(0), weights.superpoint_state_dict(0), weights.lightglue_state_dict(0)
    with torch.no_grad():
        for case in range(n_cases):
            open_variant = case % 2 == 0
            h, w = ri(40, 200), ri(40, 230)
            if not open_variant:
                h, w = h // 8 * 8, w // 8 * 8  # (the official class asserts nothing here; its eval configs use multiples of 8)
            b = ri(1, 3)
            r = ri(0, 4) if open_variant else ri(1, 4)
            border = (0, 4, 6)[ri(0, 2)] if open_variant else (2, 4)[ri(0, 1)]
            th = (0.0, 0.005, 0.02)[ri(0, 2)]
            k = (None, 40, 500)[ri(0, 2)]
            img = synthetic.synthetic_images(b, h, w, seed=1000 + case)
            if case % 3 == 0:
                img = torch.cat([img * 0.9, img, img * 0.8], 1).clamp(0, 1)
            conf = dict(nms_radius=r, remove_borders=border, detection_threshold=th)
            tag = f"{'open' if open_variant else 'official'} case {case}: {b}x{img.shape[1]}x{h}x{w} r={r} border={border} th={th} k={k}"
            try:
                if open_variant:
                    m = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": k, **conf}).eval().to(dev)
                    o = osp.extract(sd_open, img, "open", max_num_keypoints=k, **conf)
                    size = None
                else:
                    legacy = bool(ri(0, 1))
                    size = torch.tensor([[float(w - ri(0, 12)), float(h - ri(0, 10))]] * b) if case % 4 == 1 else None
                    m = superpoint.SuperPoint({"weights": "synthetic", "max_num_keypoints": k if k else -1, "legacy_sampling": legacy,
                                               **conf}).eval().to(dev)
                    o = osp.extract(sd_off, img, "official", max_num_keypoints=k if k else -1, legacy_sampling=legacy,
                                    image_size=size, **conf)
                for i in range(b):
                    data = {"image": img[i:i + 1].to(dev)}
                    if size is not None:
                        data["image_size"] = size[i:i + 1].to(dev)
                    p = m(data)
                    n_ref = len(o["keypoints"][i])
                    if p["keypoints"].shape[1] != n_ref:
                        bad.append((tag, i, "count", int(p["keypoints"].shape[1]), n_ref))
                        continue
                    if n_ref:
                        compare_keypoints(f"fuzz_{case}_{i}", p["keypoints"][0], p["keypoint_scores"][0], p["descriptors"][0],
                                          o["keypoints"][i], o["keypoint_scores"][i], o["descriptors"][i], radius=max(r, 1))
            except AssertionError as e:
                import traceback
                bad.append((tag, "assert", str(e)[:200], traceback.format_exc().splitlines()[-3:]))
    print("superpoint modules: bad", bad)
    failures = list(bad)

    bad, worst = [], 0.0
    mat = {th: lightglue.LightGlue({"weights": "synthetic", "filter_threshold": th}).eval().to(dev) for th in (0.0, 0.1, 0.3)}
    with torch.no_grad():
        for case in range(n_cases):
            b = ri(1, 3)
            m_, n_ = (ri(1, 6), ri(1, 6)) if case % 5 == 0 else (ri(8, 700), ri(8, 700))
            th = (0.0, 0.1, 0.3)[ri(0, 2)]
            size = torch.tensor([[float(ri(200, 700)), float(ri(150, 500))]] * b)
            kp0, kp1 = torch.rand((b, m_, 2), generator=g) * size[:, None], torch.rand((b, n_, 2), generator=g) * size[:, None]
            d0 = F.normalize(torch.randn((b, m_, 256), generator=g), dim=-1)
            d1 = F.normalize(torch.randn((b, n_, 256), generator=g), dim=-1)
            if case % 2:
                c = min(m_, n_)
                d1[:, :c] = F.normalize(d0[:, :c] + 0.1 * torch.randn((b, c, 256), generator=g), dim=-1)
            p = mat[th]({"keypoints0": kp0.to(dev), "keypoints1": kp1.to(dev), "descriptors0": d0.to(dev), "descriptors1": d1.to(dev),
                         "view0": {"image_size": size.to(dev)}, "view1": {"image_size": size.to(dev)}})
            o = olg.match(sd_lg, kp0, kp1, d0, d1, size, size, filter_threshold=th)
            same = torch.equal(p["matches0"].cpu(), o["matches0"]) and torch.equal(p["matches1"].cpu(), o["matches1"])
            es = (p["matching_scores0"].cpu() - o["matching_scores0"]).abs().max().item()
            la = ((p["log_assignment"].cpu() - o["log_assignment"]).abs() / (1 + o["log_assignment"].abs())).max().item()
            worst = max(worst, es, la)
            if not same:
                # a differing entry must be a near tie of the threshold or of two assignment scores
                diff = (p["matches0"].cpu() != o["matches0"]).nonzero()
                sc = torch.maximum(p["matching_scores0"].cpu(), o["matching_scores0"])[diff[:, 0], diff[:, 1]]
                near = bool(((sc - th).abs() < 1e-4).all())
                if not near:
                    bad.append(("lightglue", case, b, m_, n_, th, "matches differ", int(diff.shape[0])))
            if not es < 1e-4 or not la < 1e-4:
                bad.append(("lightglue", case, b, m_, n_, th, es, la))
    print("lightglue module: worst", worst, "bad", bad)
    return failures + bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 99) else 0)

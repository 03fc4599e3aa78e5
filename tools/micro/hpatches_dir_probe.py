"""BASELINE config 3 FROM FILES: an HPatches-structured directory of binary PPMs (synthetic images at HPatches-like
sizes, five pairs per sequence sharing image 1) -> `eval_hpatches.HPatchesPipeline.get_predictions` (read + decode on the
host, copy, resize on the GPU, pair-batched extraction and matching, records to predictions.h5) -> pairs/s for several
numbers of decode workers, then the evaluation pass (metrics + DLT on the GPU).

    python tools/micro/hpatches_dir_probe.py [n_pairs=240] [workers ...=0 4 8 16]
"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from glue_factory_colon_amd import eval_hpatches, synthetic  # noqa: E402


def write_ppm(path, img):
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(b"P6\n" + f"{w} {h}\n255\n".encode() + img.tobytes())


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    workers = [int(a) for a in sys.argv[2:]] or [0, 4, 8, 16]
    root = tempfile.mkdtemp(prefix="gfc_hp_")
    try:
        raw = synthetic.hpatches_like_host_images(n_pairs, seed=7000, pin=False, shared_view0=True)
        nbytes = 0
        for i, it in enumerate(raw):
            d = os.path.join(root, "hpatches-sequences-release", "v_" + it["scene"])
            os.makedirs(d, exist_ok=True)
            if i % 5 == 0:
                write_ppm(os.path.join(d, "1.ppm"), it["view0"]["image"].numpy())
                nbytes += it["view0"]["image"].numel()
            write_ppm(os.path.join(d, f"{i % 5 + 2}.ppm"), it["view1"]["image"].numpy())
            nbytes += it["view1"]["image"].numel()
            with open(os.path.join(d, f"H_1_{i % 5 + 2}"), "w") as f:
                f.write("1 0 0\n0 1 0\n0 0 1\n")
        del raw
        print(f"{n_pairs} pairs in {n_pairs // 5} sequences, {nbytes / 1e6:.0f} MB of PPM files under {root}", flush=True)
        model = eval_hpatches.build_model("synthetic", "synthetic", official=True).cuda()
        data = {"data_dir": os.path.join(root, "hpatches-sequences-release")}
        for w in workers:
            pipe = eval_hpatches.HPatchesPipeline(data, pair_batch=32, num_workers=w)
            rates = []
            for rep in range(2):  # the first pass also warms the file cache and the kernels' first-use set-up
                torch.cuda.synchronize()
                t = time.perf_counter()
                pred = pipe.get_predictions(os.path.join(root, f"exp_w{w}"), model, overwrite=True)
                torch.cuda.synchronize()
                rates.append(n_pairs / (time.perf_counter() - t))
            print(f"decode workers {w:2d}: {rates[1]:7.1f} pairs/s from files (first pass {rates[0]:.1f})", flush=True)
        t = time.perf_counter()
        summaries, _ = pipe.run_eval(pred)
        print(f"evaluation pass (CacheLoader + match metrics + DLT on the GPU): {n_pairs / (time.perf_counter() - t):.0f} pairs/s; "
              f"mean_num_matches {summaries['mean_num_matches']}")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()

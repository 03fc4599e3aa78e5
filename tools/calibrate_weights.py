"""Calibrate the batch-norm running statistics of the name-seeded SuperPoint weights.

Random conv weights followed by ReLU produce strongly positive-mean features; with
arbitrary BN statistics every descriptor collapses onto the same direction (cosine
similarity 0.93 between unrelated keypoints) and the detector logits barely vary.
A trained checkpoint does not have this problem because its BN statistics match its
activations.  This script gives the generated weights the same property: it pushes a
few synthetic images through the network layer by layer (plain torch CPU ops) and sets
every BN layer's running_mean / running_var to the statistics of its own input.

Output: glue-factory-colon_amd/calib/sp_calib_seed{S}.npz (a few KB of per-channel
vectors, committed; loaded by weights.py).  Run once per seed:
    python tools/calibrate_weights.py --seed 0
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "glue-factory-colon_amd")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def calibrate_open(W, S, seed, n_img=4, h=240, w=320):
    sd = W.superpoint_open_state_dict(seed, calibrated=False)
    x = S.synthetic_images(n_img, h, w, seed=4321)
    out = {}

    def block(x, p, relu=True):
        x = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=sd[p + ".conv.weight"].shape[-1] // 2)
        if relu:
            x = F.relu(x)
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        out[p + ".bn.running_mean"] = mean.numpy()
        out[p + ".bn.running_var"] = var.numpy()
        g, b = sd[p + ".bn.weight"], sd[p + ".bn.bias"]
        return (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-3) * g[None, :, None, None] \
            + b[None, :, None, None]

    for b in range(4):
        for j in range(2):
            x = block(x, f"backbone.{b}.{j}")
        if b < 3:
            x = F.max_pool2d(x, 2, 2)
    block(block(x, "detector.0"), "detector.1", relu=False)
    block(block(x, "descriptor.0"), "descriptor.1", relu=False)
    return out


def calibrate_official(W, S, seed, n_img=4, h=240, w=320):
    sd = W.superpoint_state_dict(seed, calibrated=False)
    x = S.synthetic_images(n_img, h, w, seed=4321)

    def conv(x, n, relu=True):
        x = F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], padding=sd[n + ".weight"].shape[-1] // 2)
        return F.relu(x) if relu else x

    for n in ("conv1a", "conv1b"):
        x = conv(x, n)
    x = F.max_pool2d(x, 2, 2)
    for n in ("conv2a", "conv2b"):
        x = conv(x, n)
    x = F.max_pool2d(x, 2, 2)
    for n in ("conv3a", "conv3b"):
        x = conv(x, n)
    x = F.max_pool2d(x, 2, 2)
    for n in ("conv4a", "conv4b"):
        x = conv(x, n)
    out = {}
    p = conv(conv(x, "convPa"), "convPb", relu=False)
    out["convPb.bias"] = (sd["convPb.bias"] - p.mean((0, 2, 3))).numpy()
    d = conv(conv(x, "convDa"), "convDb", relu=False)
    out["convDb.bias"] = (sd["convDb.bias"] - d.mean((0, 2, 3))).numpy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    W, S = _load("weights"), _load("synthetic")
    torch.set_grad_enabled(False)
    res = {"open/" + k: v for k, v in calibrate_open(W, S, args.seed).items()}
    res.update({"official/" + k: v for k, v in calibrate_official(W, S, args.seed).items()})
    path = os.path.join(PKG, "calib", f"sp_calib_seed{args.seed}.npz")
    np.savez_compressed(path, **{k: v.astype(np.float32) for k, v in res.items()})
    print("wrote", path, os.path.getsize(path), "bytes,", len(res), "vectors")


if __name__ == "__main__":
    sys.exit(main())

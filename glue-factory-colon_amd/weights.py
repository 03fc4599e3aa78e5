"""Deterministic, name-seeded weights for SuperPoint (open / official key layout)
and LightGlue.

There is no network in the build container or on the GPU box, so no pretrained
checkpoint can be fetched.  Parity and the benchmark therefore run on weights that
are a pure function of (tensor name, seed): the reference modules, the oracle and
the HIP path all load the very same tensors without committing 52 MB of floats.
Real checkpoints use the same key names and load through
the same `load_state_dict` of the boundary modules.

Key layouts follow the reference:
  SuperPoint-open   gluefactory/models/extractors/superpoint_open.py:61-118
  SuperPoint (off.) gluefactory_nonfree/superpoint.py:183-200
  LightGlue         gluefactory/models/matchers/lightglue.py:349-408
"""
import math
import os
import zlib
from collections import OrderedDict

import numpy as np
import torch

_CALIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "calib")


def _apply_calibration(sd, seed, variant):
    """Overwrite BN statistics / head biases with the committed calibration vectors
    (tools/calibrate_weights.py) so activations are well conditioned, as in a trained net."""
    path = os.path.join(_CALIB_DIR, f"sp_calib_seed{seed}.npz")
    if not os.path.exists(path):
        return sd
    with np.load(path) as z:
        for key in z.files:
            v, name = key.split("/", 1)
            if v == variant and name in sd and tuple(sd[name].shape) == z[key].shape:
                sd[name] = torch.from_numpy(z[key].astype(np.float32))
    return sd


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _randn(name, seed, shape, std=1.0, mean=0.0):
    return torch.randn(shape, generator=_gen(name, seed), dtype=torch.float32) * std + mean


def _rand(name, seed, shape, lo, hi):
    return torch.rand(shape, generator=_gen(name, seed), dtype=torch.float32) * (hi - lo) + lo


def _conv(sd, prefix, cin, cout, k, seed, gain=math.sqrt(2.0)):
    fan_in = cin * k * k
    sd[prefix + ".weight"] = _randn(prefix + ".weight", seed, (cout, cin, k, k), gain / math.sqrt(fan_in))
    sd[prefix + ".bias"] = _randn(prefix + ".bias", seed, (cout,), 0.05)


def _bn(sd, prefix, c, seed):
    sd[prefix + ".weight"] = _rand(prefix + ".weight", seed, (c,), 0.7, 1.3)
    sd[prefix + ".bias"] = _randn(prefix + ".bias", seed, (c,), 0.1)
    sd[prefix + ".running_mean"] = _randn(prefix + ".running_mean", seed, (c,), 0.1, 0.4)
    sd[prefix + ".running_var"] = _rand(prefix + ".running_var", seed, (c,), 0.5, 1.5)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(1, dtype=torch.long)


def superpoint_open_state_dict(seed: int = 0, channels=(64, 64, 128, 128, 256), descriptor_dim=256,
                               calibrated: bool = True):
    """84 tensors, key names of superpoint_open.py:97-118 (VGGBlock = conv, activation, bn)."""
    sd = OrderedDict()
    chans = [1, *channels[:-1]]
    for b in range(len(chans) - 1):
        cin, c = chans[b], chans[b + 1]
        for j, (ci, co) in enumerate(((cin, c), (c, c))):
            _conv(sd, f"backbone.{b}.{j}.conv", ci, co, 3, seed)
            _bn(sd, f"backbone.{b}.{j}.bn", co, seed)
    c = channels[-1]
    stride = 2 ** (len(channels) - 2)
    _conv(sd, "detector.0.conv", chans[-1], c, 3, seed)
    _bn(sd, "detector.0.bn", c, seed)
    # a larger gain on the logits makes the 65-way softmax peaky (distinct local maxima)
    _conv(sd, "detector.1.conv", c, stride * stride + 1, 1, seed, gain=4.0)
    _bn(sd, "detector.1.bn", stride * stride + 1, seed)
    _conv(sd, "descriptor.0.conv", chans[-1], c, 3, seed)
    _bn(sd, "descriptor.0.bn", c, seed)
    _conv(sd, "descriptor.1.conv", c, descriptor_dim, 1, seed, gain=1.0)
    _bn(sd, "descriptor.1.bn", descriptor_dim, seed)
    # the descriptor must not carry a constant offset, or all descriptors collapse to
    # the same direction after L2 normalisation
    sd["descriptor.1.bn.bias"] = torch.zeros(descriptor_dim)
    if calibrated and tuple(channels) == (64, 64, 128, 128, 256) and descriptor_dim == 256:
        _apply_calibration(sd, seed, "open")
    return sd


def superpoint_state_dict(seed: int = 0, descriptor_dim=256, calibrated: bool = True):
    """24 tensors, key names of gluefactory_nonfree/superpoint.py:183-200."""
    sd = OrderedDict()
    spec = [("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
            ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
            ("convPa", 128, 256, 3), ("convPb", 256, 65, 1), ("convDa", 128, 256, 3),
            ("convDb", 256, descriptor_dim, 1)]
    for name, ci, co, k in spec:
        gain = 8.0 if name == "convPb" else (1.0 if name == "convDb" else math.sqrt(2.0))
        _conv(sd, name, ci, co, k, seed, gain=gain)
    if calibrated and descriptor_dim == 256:
        _apply_calibration(sd, seed, "official")
    return sd


def _linear(sd, prefix, cin, cout, seed, gain=1.0, bias=True, bias_std=0.02):
    sd[prefix + ".weight"] = _randn(prefix + ".weight", seed, (cout, cin), gain / math.sqrt(cin))
    if bias:
        sd[prefix + ".bias"] = _randn(prefix + ".bias", seed, (cout,), bias_std)


def _ffn(sd, prefix, d, seed, out_gain):
    _linear(sd, prefix + ".0", 2 * d, 2 * d, seed)
    sd[prefix + ".1.weight"] = _rand(prefix + ".1.weight", seed, (2 * d,), 0.9, 1.1)
    sd[prefix + ".1.bias"] = _randn(prefix + ".1.bias", seed, (2 * d,), 0.02)
    _linear(sd, prefix + ".3", 2 * d, d, seed, gain=out_gain)


def lightglue_state_dict(seed: int = 0, input_dim=256, descriptor_dim=256, n_layers=9, num_heads=4,
                         residual_gain=0.15, qk_gain=2.0, assign_gain=6.0, add_scale_ori=False):
    """252 tensors (+2 for input_proj when input_dim != descriptor_dim), lightglue.py:349-408.

    Gains are chosen so that the random network behaves like a (weak) matcher on a
    shifted copy of the same image: small residual updates keep corresponding
    descriptors aligned through the 9 layers, and a large `final_proj` gain makes the
    dual-softmax assignment peaky enough for scores to pass `filter_threshold=0.1`.
    """
    d, h = descriptor_dim, num_heads
    hd = d // h
    sd = OrderedDict()
    if input_dim != d:
        _linear(sd, "input_proj", input_dim, d, seed)
    sd["posenc.Wr.weight"] = _randn("posenc.Wr.weight", seed, (hd // 2, 2), 1.0)
    if add_scale_ori:  # LearnableFourierPositionalEncoding(2 + 2, ...): columns for scale and orientation
        sd["posenc.Wr.weight"] = torch.cat([sd["posenc.Wr.weight"],
                                            _randn("posenc.Wr.weight.so", seed, (hd // 2, 2), 0.3)], 1)
    for i in range(n_layers):
        p = f"transformers.{i}.self_attn"
        _linear(sd, p + ".Wqkv", d, 3 * d, seed, gain=qk_gain)
        _linear(sd, p + ".out_proj", d, d, seed)
        _ffn(sd, p + ".ffn", d, seed, residual_gain)
        p = f"transformers.{i}.cross_attn"
        _linear(sd, p + ".to_qk", d, d, seed, gain=qk_gain)
        _linear(sd, p + ".to_v", d, d, seed)
        _linear(sd, p + ".to_out", d, d, seed)
        _ffn(sd, p + ".ffn", d, seed, residual_gain)
    for i in range(n_layers):
        p = f"log_assignment.{i}"
        _linear(sd, p + ".matchability", d, 1, seed, gain=0.5, bias_std=0.0)
        sd[p + ".matchability.bias"] = torch.full((1,), 2.0)
        # final_proj = scaled orthogonal-ish map: keeps md0.md1 large for aligned descriptors
        w = _randn(p + ".final_proj.weight", seed, (d, d), 1.0 / math.sqrt(d))
        sd[p + ".final_proj.weight"] = w * assign_gain
        sd[p + ".final_proj.bias"] = torch.zeros(d)
    for i in range(n_layers - 1):
        _linear(sd, f"token_confidence.{i}.token.0", d, 1, seed)
    return sd


def confidence_thresholds(n_layers=9):
    """Buffer of lightglue.py:555-558."""
    return torch.tensor([min(max(0.8 + 0.1 * math.exp(-4.0 * i / n_layers), 0.0), 1.0) for i in range(n_layers)],
                        dtype=torch.float32)


# per-layer mean / std of the un-biased token-confidence and matchability logits of `lightglue_state_dict(0)`
# on SuperPoint features of the synthetic images (measured once; only used to centre the heads below)
_TOKEN_MEAN = (-0.12, -0.04, -0.10, 0.32, -0.04, 0.49, -0.32, 0.52)
_MATCH_MEAN = (-0.09, 0.09, -0.01, 0.00, 0.05, -0.05, 0.00, -0.53)
_MATCH_STD = (0.06, 0.07, 0.08, 0.06, 0.09, 0.09, 0.09, 0.10)


def lightglue_adaptive_state_dict(seed: int = 0, gain: float = 8.0, prune_z: float = 0.84, **kw):
    """Variant of `lightglue_state_dict` for exercising adaptive depth / width.  With the plain random heads
    every point of a layer lands on the same side of the thresholds (all kept or all pruned).  Here the
    token-confidence and matchability heads of layers 0..7 get a larger gain and a per-layer bias that centres
    them on the decision thresholds (confidence ~0.9, matchability ~0.05), so that per layer about half of
    the points count as confident and roughly a fifth of those is pruned (`prune_z`: the pruning threshold sits that
    many standard deviations below the mean matchability logit: 0.84 -> ~20 % per layer, 1.5 -> ~7 %)."""
    sd = lightglue_state_dict(seed, **kw)
    for i in range(len(_TOKEN_MEAN)):
        k = f"token_confidence.{i}.token.0"
        if k + ".weight" in sd:
            sd[k + ".weight"] = sd[k + ".weight"] * gain
            sd[k + ".bias"] = torch.full_like(sd[k + ".bias"], 2.2 - gain * _TOKEN_MEAN[i])
        k = f"log_assignment.{i}.matchability"
        sd[k + ".weight"] = sd[k + ".weight"] * gain
        sd[k + ".bias"] = torch.full_like(sd[k + ".bias"], -2.94 - gain * _MATCH_MEAN[i] + prune_z * gain * _MATCH_STD[i])
    return sd


def disk_state_dict(seed: int = 0, desc_dim: int = 128):
    """kornia.feature.DISK key layout (kornia/feature/disk/_unets: `unet.path_down.<i>.1.{1.weight,3.weight,3.bias}`,
    `unet.path_up.<i>.conv.{...}`; index 1 = PReLU slopes, 3 = Conv2d; the first down block has no gate), name-seeded.
    He-scaled 5x5 filters, slopes around PReLU's initial 0.25; the heat-map channel gets a larger gain so that window
    NMS sees distinct maxima."""
    sd = OrderedDict()
    down = (3, 16, 32, 64, 64, 64)
    for i in range(5):
        prefix = f"unet.path_down.{i}.1"
        if i > 0:
            sd[prefix + ".1.weight"] = _rand(prefix + ".1.weight", seed, (down[i],), 0.1, 0.4)
        _conv(sd, prefix + ".3", down[i], down[i + 1], 5, seed)
    up = (64, 64, 64, desc_dim + 1)
    bot = (64,) + up
    hor = down[-2::-1]
    for i in range(4):
        prefix = f"unet.path_up.{i}.conv"
        cat = bot[i] + hor[i]
        sd[prefix + ".1.weight"] = _rand(prefix + ".1.weight", seed, (cat,), 0.1, 0.4)
        _conv(sd, prefix + ".3", cat, up[i], 5, seed)
    sd["unet.path_up.3.conv.3.weight"][desc_dim] *= 3.0
    return sd

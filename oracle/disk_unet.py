"""Oracle (test infrastructure): the DISK network on PyTorch-CPU -- the part of `kornia.feature.DISK` that
gluefactory/models/extractors/disk_kornia.py:24-47 reaches through `heatmap_and_dense_descriptors`.

kornia (>= 0.6.12, unpinned: pyproject.toml:31) is absent from the build container and from the GPU box, so this is a
restatement of kornia's published source (kornia/feature/disk/disk.py, kornia/feature/disk/_unets/unet.py and
blocks.py) -- PARITY UNPINNED: no reference run and no reference-held fixture exists for the network here.

    DISK.unet = Unet(in_features=3, size=5, down=[16, 32, 64, 64, 64], up=[64, 64, 64, desc_dim + 1])   "thin" setup:
      gate PReLU (one slope per channel), norm InstanceNorm2d (no affine, eps 1e-5), downsample avg_pool2d(2),
      upsample bilinear x2 (align_corners=False), padding = size // 2, bias, no dropout
    Conv            = Sequential(norm, gate, dropout, Conv2d)        -> parameter keys "...1.weight" (PReLU), "...3.*"
    ThinUnetDownBlock = Sequential(downsample, Conv)                 -> "unet.path_down.<i>.1.<...>"; the first block
                      has no downsample, norm or gate (NoOp modules, no parameters)
    ThinUnetUpBlock = upsample(bottom) ; cat([bottom_big, horizontal]) ; Conv   -> "unet.path_up.<i>.conv.<...>"
    heatmap_and_dense_descriptors: unet(images) -> descriptors = out[:, :desc_dim], heat-map = out[:, desc_dim:]
    (images must be divisible by 16 in both directions: four 2x poolings)

The functional form below reads a state dict with exactly those key names.
"""
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
DOWN = (16, 32, 64, 64, 64)
UP = (64, 64, 64)


def layer_table(desc_dim: int = 128):
    """[(key prefix of the Conv, cin, cout, has_norm_and_gate)] in execution order: 5 down blocks, 4 up blocks."""
    down_dims = (3,) + DOWN
    rows = [(f"unet.path_down.{i}.1", down_dims[i], down_dims[i + 1], i > 0) for i in range(5)]
    up = UP + (desc_dim + 1,)
    bot_dims = (DOWN[-1],) + up
    hor_dims = down_dims[-2::-1]
    rows += [(f"unet.path_up.{i}.conv", bot_dims[i] + hor_dims[i], up[i], True) for i in range(4)]
    return rows


def conv_block(sd: Dict[str, Tensor], prefix: str, x: Tensor, gated: bool) -> Tensor:
    """Conv = [InstanceNorm2d -> PReLU ->] Conv2d(5x5, padding 2)."""
    if gated:
        x = F.instance_norm(x, eps=1e-5)
        x = F.prelu(x, sd[prefix + ".1.weight"])
    return F.conv2d(x, sd[prefix + ".3.weight"], sd[prefix + ".3.bias"], padding=2)


def unet(sd: Dict[str, Tensor], images: Tensor) -> Tensor:
    if images.shape[-1] % 16 or images.shape[-2] % 16:
        raise ValueError(f"DISK: image size {tuple(images.shape[-2:])} is not divisible by 16")
    feats = [images]
    for i in range(5):
        x = feats[-1]
        if i > 0:
            x = F.avg_pool2d(x, 2)
        feats.append(conv_block(sd, f"unet.path_down.{i}.1", x, i > 0))
    bot = feats[-1]
    for i, hor in enumerate(feats[-2:0:-1]):  # f4, f3, f2, f1 (the image itself is not a skip connection)
        big = F.interpolate(bot, scale_factor=2, mode="bilinear", align_corners=False)
        bot = conv_block(sd, f"unet.path_up.{i}.conv", torch.cat([big, hor], dim=1), True)
    return bot


def heatmap_and_dense_descriptors(sd: Dict[str, Tensor], images: Tensor, desc_dim: int = 128) -> Tuple[Tensor, Tensor]:
    out = unet(sd, images)
    return out[:, desc_dim:], out[:, :desc_dim]

"""GPU counterpart of `gluefactory.utils.image.ImagePreprocessor` (reference gluefactory/utils/image.py:15-132): same
configuration keys, same returned dict (`image`, `scales`, `image_size`, `transform`, `original_image_size`, optional
square padding and `padding_mask`).  The resize itself -- kornia's antialiased bilinear resize in the reference -- runs
in `gfc_preprocess_resize`; it also accepts the decoded uint8 HxWxC image directly, fusing `numpy_image_to_torch`
(image.py:148-156).  Decoding files (cv2.imread) stays on the host and is not part of this module.
"""
import collections.abc as collections

import numpy as np
import torch

from . import _native as nat
from .base_model import merge

DEFAULT_CONF = {
    "resize": None,  # target edge length (or [h, w]), None for no resizing
    "edge_divisible_by": None,
    "side": "long",
    "interpolation": "bilinear",
    "align_corners": None,
    "antialias": True,
    "square_pad": False,
    "add_padding_mask": False,
}


def resize(img, size, align_corners=None, antialias=True, bgr=False):
    """img: float [C,H,W] / [B,C,H,W] on the GPU, or uint8 [H,W,C] / [H,W] / [B,H,W,C] (decoded image).
    -> float [.., C, size[0], size[1]]."""
    nat.require_cuda(img, "img")
    lib = nat.lib()
    oh, ow = int(size[0]), int(size[1])
    if img.dtype == torch.uint8:
        x = img if img.ndim != 2 else img[..., None]
        batched = x.ndim == 4
        x = (x if batched else x[None]).contiguous()
        b, h, w, c = x.shape
        u8 = 1
    else:
        batched = img.ndim == 4
        x = (img if batched else img[None]).float().contiguous()
        b, c, h, w = x.shape
        u8 = 0
    out = torch.empty((b, c, oh, ow), device=img.device, dtype=torch.float32)
    nat.check(lib.gfc_preprocess_resize(nat.ptr(x), u8, int(bool(bgr)), b, c, h, w, nat.ptr(out), oh, ow,
                                        int(bool(align_corners)), int(bool(antialias)), nat.stream_ptr(img.device)),
              "gfc_preprocess_resize")
    return out if batched else out[0]


class ImagePreprocessor:
    default_conf = DEFAULT_CONF

    def __init__(self, conf) -> None:
        unknown = set(conf or {}) - set(DEFAULT_CONF)
        if unknown:
            raise KeyError(f"unknown preprocessing keys {sorted(unknown)}")  # the reference's conf is struct (image.py:29)
        self.conf = merge(DEFAULT_CONF, dict(conf or {}))

    def __call__(self, img: torch.Tensor, interpolation=None) -> dict:
        """Resize and preprocess an image, return image and resize scale (image.py:33-72)."""
        u8 = img.dtype == torch.uint8
        if u8:
            h, w = (img.shape[-3], img.shape[-2]) if img.ndim >= 3 else img.shape
        else:
            h, w = img.shape[-2:]
        size = h, w
        if self.conf["resize"] is not None:
            interpolation = interpolation or self.conf["interpolation"]
            if interpolation != "bilinear":
                raise NotImplementedError(f"interpolation {interpolation!r}: only 'bilinear' is built on the GPU path")
            size = self.get_new_image_size(h, w)
            # kornia.resize returns its input unchanged when the size already matches
            if tuple(size) != (h, w) or u8:
                img = resize(img, size, self.conf["align_corners"], self.conf["antialias"])
        elif u8:
            img = resize(img, size, None, False)  # conversion only
        scale = torch.tensor([img.shape[-1] / w, img.shape[-2] / h], dtype=img.dtype, device=img.device)
        T = np.diag([float(scale[0]), float(scale[1]), 1])
        data = {"scales": scale, "image_size": np.array(size[::-1]), "transform": T,
                "original_image_size": np.array([w, h])}
        if self.conf["square_pad"]:
            sl = max(img.shape[-2:])
            data["image"] = torch.zeros(*img.shape[:-2], sl, sl, device=img.device, dtype=img.dtype)
            data["image"][:, : img.shape[-2], : img.shape[-1]] = img
            if self.conf["add_padding_mask"]:
                data["padding_mask"] = torch.zeros(*img.shape[:-3], 1, sl, sl, device=img.device, dtype=torch.bool)
                data["padding_mask"][:, : img.shape[-2], : img.shape[-1]] = True
        else:
            data["image"] = img
        return data

    def get_new_image_size(self, h: int, w: int):
        """Target (height, width) for `resize` = edge length on the side named by `side` (image.py:105-132):
        the named edge gets exactly `resize`, the other one the truncated aspect-preserving length; a 2-element
        `resize` is taken as (h, w) as it is; `edge_divisible_by` floors both edges to a multiple."""
        target, side = self.conf["resize"], self.conf["side"]
        if isinstance(target, collections.Iterable):
            if len(target) != 2:
                raise AssertionError("resize must be an int or (h, w)")
            return tuple(target)
        if side not in ("short", "long", "vert", "horz"):
            raise ValueError(f"side can be one of 'short', 'long', 'vert', and 'horz'. Got '{side}'")
        ratio = w / h
        landscape = ratio >= 1.0
        # which edge is pinned to `target`: the vertical one for "vert", for "short" on landscape images and for
        # "long" on portrait images
        pin_height = side == "vert" or (side == "short" and landscape) or (side == "long" and not landscape)
        size = [target, int(target * ratio)] if pin_height else [int(target / ratio), target]
        step = self.conf["edge_divisible_by"]
        if step is not None:
            size = [int(v // step * step) for v in size]
        return size

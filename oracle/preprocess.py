"""Oracle (test infrastructure): image preprocessing before the extractor, PyTorch-CPU.

Restates `ImagePreprocessor.__call__` / `get_new_image_size` / `numpy_image_to_torch` (reference
gluefactory/utils/image.py:33-72,105-132,148-156) and the resize they call.

PARITY UNPINNED for the resize: the reference calls `kornia.geometry.transform.resize` (third-party, `kornia >= 0.6.12`
unpinned in pyproject.toml:31, absent from this container and from /root/reference; `gluefactory.utils.image` itself
cannot be imported here because it imports cv2 and kornia).  `kornia_resize` below restates kornia's published
algorithm: blur only when down-scaling (`sigma = max((factor - 1)/2, 0.001)` per axis, kernel size
`int(max(2 * 2 * sigma, 3))` made odd, Gaussian `exp(-x^2 / (2 sigma^2))` normalised, separable, reflect border), then
`torch.nn.functional.interpolate(mode="bilinear", align_corners=...)`.  The interpolation half is torch's own operator,
so that half of the GPU kernel is pinned by torch; the blur parameters are pinned by nothing but this text.
"""
import numpy as np
import torch
import torch.nn.functional as F


def numpy_image_to_torch(image: np.ndarray) -> torch.Tensor:
    """image.py:148-156: HxWxC (or HxW) uint8 -> CxHxW float32 in [0, 1] (division in float64, one rounding)."""
    if image.ndim == 3:
        image = image.transpose((2, 0, 1))
    elif image.ndim == 2:
        image = image[None]
    else:
        raise ValueError(f"Not an image: {image.shape}")
    return torch.tensor(image / 255.0, dtype=torch.float)


def gaussian_kernel1d(ks: int, sigma: float) -> torch.Tensor:
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def kornia_resize(img: torch.Tensor, size, align_corners=None, antialias=True) -> torch.Tensor:
    """[..., C, H, W] float -> [..., C, size[0], size[1]]."""
    lead = img.shape[:-3]
    x = img.reshape((-1,) + img.shape[-3:])
    h, w = x.shape[-2:]
    if tuple(size) == (h, w):
        return img
    factors = (h / size[0], w / size[1])
    if antialias and max(factors) > 1:
        sig = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * sig[0], 3)), int(max(2.0 * 2 * sig[1], 3))]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        c = x.shape[1]
        ky, kx = gaussian_kernel1d(ks[0], sig[0]), gaussian_kernel1d(ks[1], sig[1])
        xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        xp = F.conv2d(xp, kx.view(1, 1, 1, -1).repeat(c, 1, 1, 1), groups=c)   # horizontal
        x = F.conv2d(xp, ky.view(1, 1, -1, 1).repeat(c, 1, 1, 1), groups=c)    # vertical
    out = F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=align_corners)
    return out.reshape(lead + out.shape[-3:])


def get_new_image_size(h, w, resize, side="long", edge_divisible_by=None):
    """image.py:105-132 restated as a table: (side, orientation) -> which edge is pinned to `resize`."""
    if isinstance(resize, (list, tuple)):
        assert len(resize) == 2
        return tuple(resize)
    if side not in ("short", "long", "vert", "horz"):
        raise ValueError(side)
    ar = w / h
    portrait = ar < 1.0
    pinned = {"vert": "h", "horz": "w", "short": "w" if portrait else "h", "long": "h" if portrait else "w"}[side]
    size = (resize, int(resize * ar)) if pinned == "h" else (int(resize / ar), resize)
    if edge_divisible_by is not None:
        size = [int(v // edge_divisible_by * edge_divisible_by) for v in size]
    return size


def preprocess(img: torch.Tensor, resize=None, side="long", edge_divisible_by=None, align_corners=None, antialias=True,
               square_pad=False, add_padding_mask=False) -> dict:
    """image.py:33-72 on a float [C,H,W] image."""
    h, w = img.shape[-2:]
    size = h, w
    if resize is not None:
        size = get_new_image_size(h, w, resize, side, edge_divisible_by)
        img = kornia_resize(img, size, align_corners, antialias)
    scale = torch.tensor([img.shape[-1] / w, img.shape[-2] / h]).to(img)
    data = {"scales": scale, "image_size": np.array(size[::-1]), "transform": np.diag([scale[0], scale[1], 1]),
            "original_image_size": np.array([w, h])}
    if square_pad:
        sl = max(img.shape[-2:])
        data["image"] = torch.zeros(*img.shape[:-2], sl, sl, dtype=img.dtype)
        data["image"][:, : img.shape[-2], : img.shape[-1]] = img
        if add_padding_mask:
            data["padding_mask"] = torch.zeros(*img.shape[:-3], 1, sl, sl, dtype=torch.bool)
            data["padding_mask"][:, : img.shape[-2], : img.shape[-1]] = True
    else:
        data["image"] = img
    return data

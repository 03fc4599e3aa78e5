"""Synthetic image pairs of the shape BASELINE.json's metric is quoted on.

SURVEY.md 8(d): white noise gives degenerate heat-maps, so images are band-limited
noise (a coarse random field bicubically up-sampled by 8 plus a little fine noise);
view1 is view0 shifted by a known (dx, dy) with border replication so that ground
truth correspondences exist.
"""
import torch
import torch.nn.functional as F


def synthetic_images(n: int, height: int = 480, width: int = 640, seed: int = 1234, device="cpu"):
    """[n,1,H,W] float32 in [0,1]; generated on CPU for reproducibility, then moved."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ch, cw = (height + 7) // 8, (width + 7) // 8
    coarse = torch.rand((n, 1, ch, cw), generator=g)
    img = F.interpolate(coarse, scale_factor=8, mode="bicubic", align_corners=False)[..., :height, :width]
    img = img + 0.1 * torch.rand((n, 1, height, width), generator=g)
    return img.clamp_(0.0, 1.0).contiguous().to(device)


def shift_image(img: torch.Tensor, dx: int, dy: int):
    """view1[y, x] = view0[y - dy, x - dx] with border replication."""
    n, c, h, w = img.shape
    ys = (torch.arange(h, device=img.device) - dy).clamp_(0, h - 1)
    xs = (torch.arange(w, device=img.device) - dx).clamp_(0, w - 1)
    return img[:, :, ys][:, :, :, xs].contiguous()


def synthetic_pairs(n_pairs: int, height: int = 480, width: int = 640, seed: int = 1234, dx: int = 16, dy: int = 8,
                    device="cpu"):
    v0 = synthetic_images(n_pairs, height, width, seed, device="cpu")
    v1 = shift_image(v0, dx, dy)
    return v0.to(device), v1.to(device)

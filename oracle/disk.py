"""Oracle (test infrastructure): the DISK extractor wrapper behind its network, on PyTorch-CPU.

Restates gluefactory/models/extractors/disk_kornia.py:29-137 (pad to /16, chunked forward, specular filter with
`keypoint_offset=0.5`, `pad_and_stack`, `+0.5`) and the three kornia functions it calls.  kornia (>= 0.6.12, unpinned:
pyproject.toml:31) is absent from the build container and from the GPU box, so `window_nms`, `heatmap_to_keypoints`
and `merge_with_descriptors` below are restatements of kornia's published source
(kornia/feature/disk/detector.py, kornia/feature/disk/structs.py) -- PARITY UNPINNED for those three: no reference
run and no reference-held fixture exists for them here.  The network itself (kornia's U-Net and its pretrained
weights) is out of reach: every function takes the heat-map / dense descriptors as inputs.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def window_nms(signal: Tensor, window_size: int = 5, cutoff: Optional[float] = 0.0) -> Tensor:
    """kornia.feature.disk.detector.nms: a pixel survives iff max_pool2d(return_indices=True) over the window centred
    on it points back at it (ties: the first maximum in the window's row-major scan), and signal > cutoff."""
    if window_size % 2 != 1:
        raise ValueError(f"window_size has to be odd, got {window_size}")
    _, ixs = F.max_pool2d(signal, kernel_size=window_size, stride=1, padding=window_size // 2, return_indices=True)
    h, w = signal.shape[1:]
    coords = torch.arange(h * w, device=signal.device).reshape(1, h, w)
    keep = ixs == coords
    return keep if cutoff is None else keep & (signal > cutoff)


def heatmap_to_keypoints(heatmap: Tensor, n: Optional[int] = None, window_size: int = 5,
                         score_threshold: float = 0.0) -> List[Tuple[Tensor, Tensor]]:
    """kornia.feature.disk.detector.heatmap_to_keypoints.  heatmap [B,1,H,W] -> per image (xy int64 [N,2], score [N])
    in row-major order.  With n: keep scores strictly above the (n+1)-th largest (kthvalue of the negated scores at
    min(n+1, count) -- with count <= n this drops the minimum), then the first n."""
    heatmap = heatmap.squeeze(1)
    nmsed = window_nms(heatmap, window_size=window_size, cutoff=score_threshold)
    out = []
    for b in range(heatmap.shape[0]):
        yx = nmsed[b].nonzero(as_tuple=False)
        logp = heatmap[b][nmsed[b]]
        xy = yx.flip((1,))
        if n is not None:
            if logp.numel() == 0:  # torch.kthvalue raises on an empty tensor: nothing to keep
                out.append((xy, logp))
                continue
            n_ = min(n + 1, logp.numel())
            minus_threshold, _ = torch.kthvalue(-logp, n_)
            mask = logp > -minus_threshold
            xy, logp = xy[mask][:n], logp[mask][:n]
        out.append((xy, logp))
    return out


def merge_with_descriptors(xy: Tensor, dense: Tensor) -> Tensor:
    """kornia Keypoints.merge_with_descriptors: dense [D,H,W] read at the integer pixel, F.normalize over D."""
    desc = dense[:, xy[:, 1], xy[:, 0]].T
    return F.normalize(desc, dim=-1)


def filter_specular(kp: Tensor, mask: Optional[Tensor], *values, image_size=None, keypoint_offset: float = 0.5):
    """gluefactory/models/extractors/utils.py:4-42."""
    if mask is None or kp.numel() == 0:
        return (kp, *values)
    if image_size is not None:
        w, h = image_size
        mask = mask[..., : int(h), : int(w)]
    if mask.ndim == 3:
        mask = mask.squeeze(0)
    mask = mask.to(torch.bool)
    h, w = mask.shape[-2:]
    xy = kp - keypoint_offset
    x0, x1 = torch.floor(xy[:, 0]).long(), torch.ceil(xy[:, 0]).long()
    y0, y1 = torch.floor(xy[:, 1]).long(), torch.ceil(xy[:, 1]).long()
    inside = (x0 >= 0) & (x1 < w) & (y0 >= 0) & (y1 < h)
    keep = torch.zeros_like(inside)
    if inside.any():
        keep[inside] = (mask[y0[inside], x0[inside]] & mask[y0[inside], x1[inside]] & mask[y1[inside], x0[inside]]
                        & mask[y1[inside], x1[inside]])
    return (kp[keep], *[None if v is None else v[keep] for v in values])


def extract(dense_fn, image: Tensor, max_num_keypoints: Optional[int] = None, nms_window_size: int = 5,
            detection_threshold: float = 0.0, pad_if_not_divisible: bool = True, chunk: int = 4,
            specular_mask: Optional[Tensor] = None, image_size: Optional[Tensor] = None):
    """disk_kornia.py:55-137 without force_num_keypoints (random padding).  dense_fn(images [b,3,H',W']) ->
    (heatmaps [b,1,H',W'], descriptors [b,D,H',W']).  Returns lists per image: keypoints (+0.5), scores, descriptors."""
    kps, scs, des = [], [], []
    for i in range(0, image.shape[0], chunk):
        x = image[i:i + chunk]
        h, w = x.shape[2:]
        if pad_if_not_divisible:  # disk_kornia.py:31-35 (and kornia's own forward)
            pd_h = 16 - h % 16 if h % 16 > 0 else 0
            pd_w = 16 - w % 16 if w % 16 > 0 else 0
            x = F.pad(x, (0, pd_w, 0, pd_h), value=0.0)
        heat, dense = dense_fn(x)
        heat, dense = heat[..., :h, :w], dense[..., :h, :w]
        for j, (xy, sc) in enumerate(heatmap_to_keypoints(heat, max_num_keypoints, nms_window_size, detection_threshold)):
            d = merge_with_descriptors(xy, dense[j])
            k = xy.float()
            if specular_mask is not None:  # disk_kornia.py:84-107
                isz = None if image_size is None else image_size[i + j]
                k, sc, d = filter_specular(k + 0.5, specular_mask[i + j], sc, d, image_size=isz, keypoint_offset=0.5)
                k = k - 0.5
            kps.append(k + 0.5)
            scs.append(sc)
            des.append(d)
    return kps, scs, des

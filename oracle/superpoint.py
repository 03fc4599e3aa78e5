"""Oracle (test infrastructure): SuperPoint feature extraction on PyTorch-CPU fp32.

Two arithmetic variants share the detection / selection / sampling tail:
  * "open"     : conv -> ReLU -> BatchNorm(eps=1e-3) blocks
                 (gluefactory/models/extractors/superpoint_open.py:61-118,126-232)
  * "official" : conv -> ReLU, no BN, legacy descriptor sampling by default
                 (gluefactory_nonfree/superpoint.py:183-200,206-379)
Weights are passed as a state dict with the reference's key names.  Nothing here is
used by the product path.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

OPEN_DEFAULTS = dict(nms_radius=4, max_num_keypoints=None, detection_threshold=0.005, remove_borders=4,
                     channels=(64, 64, 128, 128, 256))
OFFICIAL_DEFAULTS = dict(nms_radius=4, max_num_keypoints=-1, detection_threshold=0.005, remove_borders=4,
                         legacy_sampling=True)


def to_gray(image: Tensor) -> Tensor:
    """superpoint_open.py:128-130 / superpoint.py:208-210: weighted RGB sum when C == 3."""
    if image.shape[1] == 3:
        w = image.new_tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1)
        image = (image * w).sum(1, keepdim=True)
    return image


# ----------------------------------------------------------------------------- encoders
def _block_open(x: Tensor, sd: Dict[str, Tensor], prefix: str, relu: bool = True) -> Tensor:
    """One VGGBlock in eval mode: conv, optional ReLU, then BN with running statistics
    (superpoint_open.py:61-77)."""
    w = sd[prefix + ".conv.weight"]
    x = F.conv2d(x, w, sd[prefix + ".conv.bias"], stride=1, padding=(w.shape[-1] - 1) // 2)
    if relu:
        x = F.relu(x)
    return F.batch_norm(x, sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"],
                        sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"], training=False, eps=0.001)


def dense_open(sd: Dict[str, Tensor], image: Tensor) -> Tuple[Tensor, Tensor]:
    """image [B,1|3,H,W] -> (heat-map [B,H,W] before NMS, L2-normalised dense descriptors [B,256,H/8,W/8]).
    superpoint_open.py:100-118 (layers), :132-144 (forward)."""
    x = to_gray(image)
    n_blocks = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("backbone."))
    for b in range(n_blocks):
        x = _block_open(x, sd, f"backbone.{b}.0")
        x = _block_open(x, sd, f"backbone.{b}.1")
        if b < n_blocks - 1:
            x = F.max_pool2d(x, kernel_size=2, stride=2)
    desc = _block_open(_block_open(x, sd, "descriptor.0"), sd, "descriptor.1", relu=False)
    desc = F.normalize(desc, p=2, dim=1)
    logits = _block_open(_block_open(x, sd, "detector.0"), sd, "detector.1", relu=False)
    return logits_to_heatmap(logits), desc


def dense_official(sd: Dict[str, Tensor], image: Tensor) -> Tuple[Tensor, Tensor]:
    """Official SuperPoint encoder + heads (gluefactory_nonfree/superpoint.py:183-200,212-241):
    ReLU after every conv except the two 1x1 heads, 2x2 max-pool after 1b, 2b, 3b."""
    x = to_gray(image)

    def conv(x, name, relu=True):
        w = sd[name + ".weight"]
        x = F.conv2d(x, w, sd[name + ".bias"], stride=1, padding=(w.shape[-1] - 1) // 2)
        return F.relu(x) if relu else x

    for stage in ("1", "2", "3", "4"):
        x = conv(conv(x, f"conv{stage}a"), f"conv{stage}b")
        if stage != "4":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
    logits = conv(conv(x, "convPa"), "convPb", relu=False)
    desc = conv(conv(x, "convDa"), "convDb", relu=False)
    return logits_to_heatmap(logits), F.normalize(desc, p=2, dim=1)


def logits_to_heatmap(logits: Tensor) -> Tensor:
    """[B,s*s+1,h,w] -> [B,h*s,w*s]: softmax over channels, drop the dustbin, depth-to-space with
    S[b, s*y+i, s*x+j] = P[b, s*i+j, y, x]   (superpoint_open.py:139-144; superpoint.py:231-235)."""
    p = torch.softmax(logits, dim=1)[:, :-1]
    b, c, h, w = p.shape
    s = int(round(c ** 0.5))
    p = p.view(b, s, s, h, w).permute(0, 3, 1, 4, 2)  # b, y, i, x, j
    return p.reshape(b, h * s, w * s)


# --------------------------------------------------------------------------------- NMS
def nms(scores: Tensor, radius: int) -> Tensor:
    """Max-pool non-maximum suppression with two recovery rounds
    (superpoint_open.py:36-51 == superpoint.py:63-83).  scores [B,H,W]."""
    assert radius >= 0
    k = 2 * radius + 1

    def pool(t):
        return F.max_pool2d(t, kernel_size=k, stride=1, padding=radius)

    zero = torch.zeros_like(scores)
    keep = scores == pool(scores)
    for _ in range(2):
        near_kept = pool(keep.float()) > 0
        masked = torch.where(near_kept, zero, scores)
        fresh = (masked == pool(masked)) & ~near_kept
        keep = keep | fresh
    return torch.where(keep, scores, zero)


def kill_borders(scores: Tensor, pad: int, image_size: Optional[Tensor] = None) -> Tensor:
    """Write -1 into the outer `pad` rows / columns of a copy of scores [B,H,W].
    Open variant: all four sides from the tensor shape (superpoint_open.py:148-154).
    Official variant with image_size [B,2]=(w,h): right/bottom measured from the true
    image extent (superpoint.py:249-260)."""
    s = scores.clone()
    if not pad:
        return s
    s[:, :pad] = -1
    s[:, :, :pad] = -1
    if image_size is None:
        s[:, -pad:] = -1
        s[:, :, -pad:] = -1
    else:
        for i in range(s.shape[0]):
            w, h = int(image_size[i, 0].item()), int(image_size[i, 1].item())
            s[i, h - pad:] = -1
            s[i, :, w - pad:] = -1
    return s


def select_keypoints(scores: Tensor, threshold: float, k: Optional[int]) -> Tuple[Tensor, Tensor]:
    """One image [H,W] -> (xy [N,2] float32, score [N]).  Candidates are the pixels with
    score > threshold in row-major order; if there are more than k, the k best in
    descending score order, otherwise all of them in row-major order
    (superpoint_open.py:156-192,54-58; superpoint.py:262-300,86-90)."""
    ys, xs = torch.where(scores > threshold)
    vals = scores[ys, xs]
    xy = torch.stack([xs, ys], -1).float()
    if k is not None and k >= 0 and k < len(vals):
        vals, idx = torch.topk(vals, k, dim=0, sorted=True)
        xy = xy[idx]
    return xy, vals


def soft_argmax_refinement(xy: Tensor, scores: Tensor, radius: int) -> Tensor:
    """gluefactory_nonfree/superpoint.py:100-116 for one image: xy [N,2] integer-valued (x, y), scores [H,W] the
    dense heat-map -> xy + score-weighted mean offset in the (2r+1)^2 window (zero padding)."""
    width = 2 * radius + 1
    s4 = scores[None, None]
    sum_ = F.avg_pool2d(s4, width, 1, radius, divisor_override=1)
    ar = torch.arange(-radius, radius + 1).to(scores)
    kernel_x = ar[None].expand(width, -1)[None, None]
    dx = F.conv2d(s4, kernel_x, padding=radius)
    dy = F.conv2d(s4, kernel_x.transpose(2, 3), padding=radius)
    dxdy = torch.stack([dx[0, 0], dy[0, 0]], -1) / sum_[0, 0, :, :, None]
    idx = xy.long()
    return xy.float() + dxdy[idx[:, 1], idx[:, 0]]


def filter_keypoints_by_specular_mask(keypoints: Tensor, specular_mask: Optional[Tensor], *values, image_size=None,
                                      keypoint_offset: float = 0.5):
    """extractors/utils.py:4-42 (the Endomapper addition of this reference): keep the key points whose
    floor/ceil(kp - offset) pixels are inside the mask (cropped to image_size = (w, h)) and set on all four."""
    if specular_mask is None or keypoints.numel() == 0:
        return (keypoints, *values)
    mask = specular_mask
    if image_size is not None:
        w, h = image_size
        mask = mask[..., : int(h), : int(w)]
    if mask.ndim == 3:
        mask = mask.squeeze(0)
    mask = mask.to(torch.bool)
    h, w = mask.shape[-2:]
    xy = keypoints - keypoint_offset
    x0, x1 = torch.floor(xy[:, 0]).long(), torch.ceil(xy[:, 0]).long()
    y0, y1 = torch.floor(xy[:, 1]).long(), torch.ceil(xy[:, 1]).long()
    inside = (x0 >= 0) & (x1 < w) & (y0 >= 0) & (y1 < h)
    keep = torch.zeros_like(inside)
    if inside.any():
        keep[inside] = (mask[y0[inside], x0[inside]] & mask[y0[inside], x1[inside]]
                        & mask[y1[inside], x0[inside]] & mask[y1[inside], x1[inside]])
    return (keypoints[keep], *[None if v is None else v[keep] for v in values])


# ---------------------------------------------------------------------------- sampling
def sample_descriptors(keypoints: Tensor, dense: Tensor, s: int = 8, mode: str = "open") -> Tensor:
    """Bilinear read of dense descriptors [B,C,h,w] at integer-pixel keypoints [B,N,2] (x,y),
    then L2 normalisation -> [B,N,C].
      "open"   : g = (kp + .5) / ([w,h]*s) * 2 - 1, align_corners=False  (superpoint_open.py:22-33)
      "legacy" : g = (kp - s/2 + .5) / ([w,h]*s - s/2 - .5) * 2 - 1, align_corners=True (superpoint.py:120-138)
      "fixed"  : g = kp / ([w,h]*s) * 2 - 1, align_corners=False       (superpoint.py:141-152)"""
    b, c, h, w = dense.shape
    kp = keypoints.clone()
    if mode == "open":
        g = (kp + 0.5) / (kp.new_tensor([w, h]) * s)
        align = False
    elif mode == "legacy":
        g = (kp - s / 2 + 0.5) / kp.new_tensor([w * s - s / 2 - 0.5, h * s - s / 2 - 0.5])[None]
        align = True
    elif mode == "fixed":
        g = kp / (kp.new_tensor([w, h]) * s)
        align = False
    else:
        raise ValueError(mode)
    g = g * 2 - 1
    out = F.grid_sample(dense, g.view(b, 1, -1, 2), mode="bilinear", align_corners=align)
    out = F.normalize(out.reshape(b, c, -1), p=2, dim=1)
    return out.transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------- forward
def extract(sd: Dict[str, Tensor], image: Tensor, variant: str = "open", nms_radius: int = 4,
            max_num_keypoints: Optional[int] = None, detection_threshold: float = 0.005, remove_borders: int = 4,
            legacy_sampling: bool = True, image_size: Optional[Tensor] = None,
            specular_mask: Optional[Tensor] = None, refinement_radius: int = 0) -> Dict[str, object]:
    """Full extractor.  Returns per-image lists (ragged) plus the intermediates the parity
    tests compare stage by stage:
      heatmap [B,H,W], nms [B,H,W] (after border kill), keypoints: list of [N_i,2] (x+.5,y+.5),
      keypoint_scores: list of [N_i], descriptors: list of [N_i,C], dense_descriptors [B,C,h,w]."""
    with torch.no_grad():
        if variant == "open":
            heat, dense = dense_open(sd, image)
            mode = "open"
            size = None
        else:
            heat, dense = dense_official(sd, image)
            mode = "legacy" if legacy_sampling else "fixed"
            size = image_size
            if max_num_keypoints is not None and max_num_keypoints <= 0:
                max_num_keypoints = None
        suppressed = kill_borders(nms(heat, nms_radius), remove_borders, size)
        kpts: List[Tensor] = []
        scores: List[Tensor] = []
        descs: List[Tensor] = []
        for i in range(image.shape[0]):
            isz = None if image_size is None else image_size[i]
            if specular_mask is not None and variant == "open":
                # superpoint_open.py:156-190: candidates -> specular filter -> top-k
                xy, sc = select_keypoints(suppressed[i], detection_threshold, None)
                xy, sc = filter_keypoints_by_specular_mask(xy, specular_mask[i], sc, image_size=isz, keypoint_offset=0.0)
                if max_num_keypoints is not None and max_num_keypoints < len(sc):
                    sc, idx = torch.topk(sc, max_num_keypoints, dim=0, sorted=True)
                    xy = xy[idx]
            else:
                xy, sc = select_keypoints(suppressed[i], detection_threshold, max_num_keypoints)
                if refinement_radius > 0 and variant != "open":  # superpoint.py:302-305
                    xy = soft_argmax_refinement(xy, heat[i], refinement_radius)
                if specular_mask is not None:  # superpoint.py:310-328: top-k -> specular filter
                    xy, sc = filter_keypoints_by_specular_mask(xy, specular_mask[i], sc, image_size=isz,
                                                               keypoint_offset=0.0)
            d = sample_descriptors(xy[None], dense[i:i + 1], 8, mode)[0]
            kpts.append(xy + 0.5)
            scores.append(sc)
            descs.append(d)
    return {"heatmap": heat, "nms": suppressed, "keypoints": kpts, "keypoint_scores": scores,
            "descriptors": descs, "dense_descriptors": dense}

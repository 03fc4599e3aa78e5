"""Oracle (test infrastructure): homography match metrics of the HPatches evaluation
(SURVEY.md 8f rank 2), PyTorch-CPU.

Restates gluefactory/eval/utils.py:141-185 (`eval_matches_homography`),
gluefactory/geometry/homography.py:161-180,314-323 (`warp_points_torch`, `sym_homography_error`)
and gluefactory/geometry/gt_generation.py:730-801 (`gt_matches_from_homography`).

Pinning: `warp_points` / `sym_homography_error` / `homography_corner_error` are checked against golden
vectors produced by the reference's own `geometry.homography` (importable in the build container);
`eval_matches_homography` against the known-answer cases of the reference's tests/test_eval_utils.py:30-88.
`gt_matches_from_homography` lives in a module whose import needs kornia (absent): it is restated from the
source text and its outputs are NOT pinned by a reference run (parity unpinned for the two gt_match_* metrics).
"""
import torch

UNMATCHED, IGNORE = -1, -2


def to_h(p):
    return torch.cat([p, torch.ones_like(p[..., :1])], -1)


def from_h(p, eps=0.0):
    return p[..., :-1] / (p[..., -1:] + eps)


def warp_points(points, H, inverse=True):
    """homography.py:161-180: multiply by H (or its inverse), divide by (w + 1e-5)."""
    Hm = (torch.inverse(H) if inverse else H).transpose(-2, -1)
    return from_h(torch.einsum("...nj,...ji->...ni", to_h(points), Hm), eps=1e-5)


def sym_homography_error(kpts0, kpts1, H):
    """homography.py:314-323: mean of the two transfer distances (inverse through pinverse)."""
    k01 = from_h(to_h(kpts0) @ H.transpose(-1, -2))
    d01 = ((k01 - kpts1) ** 2).sum(-1).sqrt()
    k10 = from_h(to_h(kpts1) @ torch.pinverse(H.transpose(-1, -2)))
    d10 = ((k10 - kpts0) ** 2).sum(-1).sqrt()
    return (d01 + d10) / 2.0


def homography_corner_error(T, T_gt, image_size):
    """homography.py:337-344."""
    W, H = image_size[..., 0], image_size[..., 1]
    c0 = torch.tensor([[0, 0], [W, 0], [W, H], [0, H]], dtype=T.dtype)
    c1_gt = from_h(to_h(c0) @ T_gt.transpose(-1, -2))
    c1 = from_h(to_h(c0) @ T.transpose(-1, -2))
    return torch.sqrt(((c1 - c1_gt) ** 2).sum(-1)).mean(-1)


def gt_matches_from_homography(kp0, kp1, H, pos_th=3.0, neg_th=6.0):
    """gt_generation.py:730-801 (no validity masks).  kp0 [B,M,2], kp1 [B,N,2], H [B,3,3] ->
    matches0 [B,M], matches1 [B,N] with -1 = unmatched, -2 = ignore."""
    if kp0.shape[1] == 0 or kp1.shape[1] == 0:
        return (-torch.ones_like(kp0[:, :, 0]).long(), -torch.ones_like(kp1[:, :, 0]).long())
    k01 = warp_points(kp0, H, inverse=False)
    k10 = warp_points(kp1, H, inverse=True)
    d0 = ((k01.unsqueeze(-2) - kp1.unsqueeze(-3)) ** 2).sum(-1)
    d1 = ((kp0.unsqueeze(-2) - k10.unsqueeze(-3)) ** 2).sum(-1)
    dist = torch.max(d0, d1)
    min0, min1 = dist.min(-1).indices, dist.min(-2).indices
    is0 = torch.zeros_like(dist, dtype=torch.bool).scatter_(-1, min0.unsqueeze(-1), True)
    is1 = torch.zeros_like(dist, dtype=torch.bool).scatter_(-2, min1.unsqueeze(-2), True)
    positive = is0 & is1 & (dist < pos_th ** 2)
    neg0 = d0.min(-1).values > neg_th ** 2
    neg1 = d1.min(-2).values > neg_th ** 2
    m0 = torch.where(positive.any(-1), min0, min0.new_tensor(IGNORE))
    m1 = torch.where(positive.any(-2), min1, min1.new_tensor(IGNORE))
    m0 = torch.where(neg0, m0.new_tensor(UNMATCHED), m0)
    m1 = torch.where(neg1, m1.new_tensor(UNMATCHED), m1)
    return m0, m1


def eval_matches_homography(H_gt, kp0, kp1, m0):
    """eval/utils.py:141-185 for ONE pair: kp0 [M,2], kp1 [N,2], m0 [M] -> dict of floats."""
    ok = m0 > -1
    pts0, pts1 = kp0[ok], kp1[m0[ok]]
    err = sym_homography_error(pts0, pts1, H_gt)
    res = {"prec@1px": (err < 1).float().mean().nan_to_num().item(),
           "prec@3px": (err < 3).float().mean().nan_to_num().item(),
           "num_matches": int(pts0.shape[0]), "num_keypoints": (kp0.shape[0] + kp1.shape[0]) / 2.0}
    gt0, _ = gt_matches_from_homography(kp0[None], kp1[None], H_gt[None], pos_th=3.0, neg_th=3.0)
    m, g = m0[None], gt0
    mask = (g > -1).float()
    res["gt_match_recall@3px"] = (((m == g) * mask).sum(1) / (1e-8 + mask.sum(1)))[0].item()
    mask = ((m > -1) & (g >= -1)).float()
    res["gt_match_precision@3px"] = (((m == g) * mask).sum(1) / (1e-8 + mask.sum(1)))[0].item()
    return res

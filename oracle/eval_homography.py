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


# ---- weighted DLT homography + corner error -------------------------------------------
# The reference calls kornia.geometry.homography.find_homography_dlt (third-party, `kornia >= 0.6.12`, unpinned in
# pyproject.toml:31; ABSENT from this container and from /root/reference) at gluefactory/eval/utils.py:276-302.
# PARITY UNPINNED for the solver: the function below restates kornia's published algorithm (Hartley-normalised DLT:
# normalize_points -> 2 rows per correspondence -> A^T diag(w) A -> right singular vector of the smallest singular
# value -> de-normalise -> divide by (H[2,2] + 1e-8)); it is pinned only by properties (exact correspondences
# recover H_gt; the solution is the minimiser of the weighted algebraic residual) and, for the corner error it feeds,
# by the reference-generated golden vectors of `homography_corner_error` above.


def normalize_points(points, eps=1e-8):
    """kornia.geometry.epipolar.normalize_points: centroid to origin, mean distance sqrt(2).  [B,N,2] -> pts, T [B,3,3]."""
    mean = points.mean(1, keepdim=True)
    scale = (points - mean).norm(dim=-1, p=2).mean(-1)
    scale = (2.0 ** 0.5) / (scale + eps)
    b = points.shape[0]
    T = torch.zeros((b, 3, 3), dtype=points.dtype)
    T[:, 0, 0] = scale
    T[:, 1, 1] = scale
    T[:, 0, 2] = -scale * mean[:, 0, 0]
    T[:, 1, 2] = -scale * mean[:, 0, 1]
    T[:, 2, 2] = 1.0
    pn = from_h(to_h(points) @ T.transpose(-1, -2))
    return pn, T


def find_homography_dlt(points1, points2, weights=None):
    """Weighted normalised DLT, [B,N,2] x [B,N,2] (x [B,N]) -> H [B,3,3] with points2 ~ H points1.
    Raises AssertionError for fewer than 4 correspondences (the caller maps that to H = inf, eval/utils.py:291-292)."""
    assert points1.shape == points2.shape and points1.shape[1] >= 4
    p1, T1 = normalize_points(points1)
    p2, T2 = normalize_points(points2)
    x1, y1 = p1[..., 0:1], p1[..., 1:2]
    x2, y2 = p2[..., 0:1], p2[..., 1:2]
    one, zero = torch.ones_like(x1), torch.zeros_like(x1)
    ax = torch.cat([zero, zero, zero, -x1, -y1, -one, y2 * x1, y2 * y1, y2], -1)
    ay = torch.cat([x1, y1, one, zero, zero, zero, -x2 * x1, -x2 * y1, -x2], -1)
    A = torch.cat([ax, ay], -1).reshape(points1.shape[0], -1, 9)
    if weights is None:
        AtA = A.transpose(-2, -1) @ A
    else:
        w = weights.unsqueeze(-1).repeat(1, 1, 2).reshape(points1.shape[0], -1)
        AtA = A.transpose(-2, -1) @ (w.unsqueeze(-1) * A)
    _, _, Vh = torch.linalg.svd(AtA)
    Hn = Vh[:, -1, :].reshape(-1, 3, 3)
    H = torch.inverse(T2) @ (Hn @ T1)
    return H / (H[..., -1:, -1:] + 1e-8)


def eval_homography_dlt(H_gt, kp0, kp1, m0, scores0, image_size0):
    """eval/utils.py:276-302 for one pair: matched points weighted by their matching scores -> H_dlt ->
    mean corner distance to H_gt; inf when the estimate is not finite or there are fewer than 4 matches."""
    valid = m0 > -1
    pts0, pts1, sc = kp0[valid], kp1[m0[valid]], scores0[valid].to(kp0)
    try:
        H = find_homography_dlt(pts0[None], pts1[None], sc[None])[0]
    except AssertionError:
        H = torch.full((3, 3), float("inf"))
    if not torch.isfinite(H).all():
        return H, float("inf")
    err = homography_corner_error(H, H_gt, image_size0)
    return H, (float(err) if torch.isfinite(err) else float("inf"))

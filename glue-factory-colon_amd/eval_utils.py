"""HPatches match metrics on the GPU -- counterpart of `gluefactory.eval.utils.eval_matches_homography`
(reference gluefactory/eval/utils.py:141-185).  Same arguments, same result keys; the arithmetic runs in
`gfc_eval_matches_homography` (one workgroup per pair) instead of materialising the M x N distance matrix
with torch ops.  RANSAC-based estimators (opencv / poselib) stay CPU libraries and are out of scope.
"""
import torch

from . import _native as nat

RESULT_KEYS = ("prec@1px", "prec@3px", "num_matches", "num_keypoints", "gt_match_recall@3px",
               "gt_match_precision@3px")


def match_metrics(H_0to1, kp0, kp1, matches0, pos_th=3.0, neg_th=3.0, return_gt=False):
    """Batched tensors on the device: H [B,3,3], kp0 [B,M,2], kp1 [B,N,2], matches0 [B,M] -> [B,6]
    (RESULT_KEYS order) and optionally the ground-truth matches [B,M] (-1 unmatched, -2 ignore)."""
    nat.require_cuda(kp0, "keypoints0")
    lib = nat.lib()
    dev = kp0.device
    b, m, n = kp0.shape[0], kp0.shape[1], kp1.shape[1]
    H = H_0to1.to(device=dev, dtype=torch.float32).reshape(b, 3, 3).contiguous()
    Hinv = torch.linalg.inv(H.double()).float().contiguous()  # 3x3 plumbing
    k0, k1 = kp0.float().contiguous(), kp1.float().contiguous()
    m0 = matches0.to(torch.long).contiguous()
    out = torch.empty((b, 6), device=dev, dtype=torch.float32)
    gt = torch.empty((b, m), device=dev, dtype=torch.long) if return_gt else None
    nat.check(lib.gfc_eval_matches_homography(nat.ptr(k0), nat.ptr(k1), nat.ptr(m0), nat.ptr(H), nat.ptr(Hinv), b, m,
                                              n, float(pos_th), float(neg_th), nat.ptr(out), nat.ptr(gt),
                                              nat.stream_ptr(dev)), "gfc_eval_matches_homography")
    return (out, gt) if return_gt else out


def eval_matches_homography(data: dict, pred: dict) -> dict:
    """Drop-in for gluefactory.eval.utils.eval_matches_homography: un-batched inputs give floats,
    batched inputs (H_0to1.ndim > 2) lists per item (eval_per_batch_item, eval/utils.py:35-50)."""
    for key in ("H_0to1",):
        assert key in data, f"Missing key {key} in data"
    for key in ("keypoints0", "keypoints1", "matches0", "matching_scores0"):
        assert key in pred, f"Missing key {key} in data"
    H = data["H_0to1"]
    batched = H.ndim > 2
    kp0, kp1, m0 = pred["keypoints0"], pred["keypoints1"], pred["matches0"]
    if not batched:
        H, kp0, kp1, m0 = H[None], kp0[None], kp1[None], m0[None]
    res = match_metrics(H, kp0, kp1, m0).cpu()
    out = {}
    for i, key in enumerate(RESULT_KEYS):
        col = res[:, i]
        vals = [int(v) if key == "num_matches" else float(v) for v in col.tolist()]
        out[key] = vals if batched else vals[0]
    return out


def homography_dlt(H_0to1, kp0, kp1, matches0, scores0, image_size0):
    """Batched device tensors -> (H_dlt [B,3,3], corner error [B]); +inf where a pair has < 4 matches."""
    nat.require_cuda(kp0, "keypoints0")
    lib = nat.lib()
    dev = kp0.device
    b, m, n = kp0.shape[0], kp0.shape[1], kp1.shape[1]
    H = H_0to1.to(device=dev, dtype=torch.float32).reshape(b, 9).contiguous()
    size = image_size0.to(device=dev, dtype=torch.float32).reshape(b, 2).contiguous()
    k0, k1 = kp0.float().contiguous(), kp1.float().contiguous()
    m0 = matches0.to(torch.long).contiguous()
    sc = scores0.to(device=dev, dtype=torch.float32).contiguous()
    Hout = torch.empty((b, 3, 3), device=dev, dtype=torch.float32)
    err = torch.empty((b,), device=dev, dtype=torch.float32)
    nat.check(lib.gfc_eval_homography_dlt(nat.ptr(k0), nat.ptr(k1), nat.ptr(m0), nat.ptr(sc), nat.ptr(H), nat.ptr(size),
                                          b, m, n, nat.ptr(Hout), nat.ptr(err), nat.stream_ptr(dev)),
              "gfc_eval_homography_dlt")
    return Hout, err


def eval_homography_dlt(data: dict, pred: dict) -> dict:
    """Drop-in for gluefactory.eval.utils.eval_homography_dlt (eval/utils.py:276-302): {"H_error_dlt": float}
    (a list per item for batched input).  The weights are the matching scores, as in the reference."""
    assert "H_0to1" in data, "Missing key H_0to1 in data"
    for key in ("keypoints0", "keypoints1", "matches0", "matching_scores0"):
        assert key in pred, f"Missing key {key} in data"
    H = data["H_0to1"]
    batched = H.ndim > 2
    kp0, kp1, m0, s0 = pred["keypoints0"], pred["keypoints1"], pred["matches0"], pred["matching_scores0"]
    size = data["view0"]["image_size"]
    if not batched:
        H, kp0, kp1, m0, s0, size = H[None], kp0[None], kp1[None], m0[None], s0[None], size[None]
    _, err = homography_dlt(H, kp0, kp1, m0, s0, size)
    vals = [float(v) for v in err.cpu().tolist()]
    return {"H_error_dlt": vals if batched else vals[0]}

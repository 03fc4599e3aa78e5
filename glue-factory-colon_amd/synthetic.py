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


HPATCHES_LIKE_SHAPES = [(480, 640), (480, 613), (640, 480), (725, 480), (480, 656)]  # (h, w): short side 480


def hpatches_shaped_pairs(n_pairs: int, seed: int = 4000, device="cpu", per_sequence: int = 5, shared_view0: bool = False):
    """`n_pairs` loader items shaped like the HPatches evaluation list (BASELINE config 3; datasets/hpatches.py:94-112:
    RGB, short side resized to 480, arbitrary long side, batch 1, `scales` = new / original size): sequences of
    `per_sequence` pairs share view 0's image shape (the sequence's reference image), view 1's shape changes from item
    to item.  Both views are crops of one band-limited canvas displaced by (24, 16) pixels, so that true
    correspondences exist.  shared_view0: like the real list (datasets/hpatches.py:98-99 reads image 1 of the sequence as
    view 0 of every one of its five pairs) all pairs of a sequence carry THE SAME view-0 image, and their view 1 is
    another crop of the sequence's canvas; items then also carry `scene` (the sequence name, as the reference's items do)."""
    items = []
    for i in range(n_pairs):
        s0 = HPATCHES_LIKE_SHAPES[(i // per_sequence) % len(HPATCHES_LIKE_SHAPES)]
        s1 = HPATCHES_LIKE_SHAPES[(i * 2 + 1) % len(HPATCHES_LIKE_SHAPES)]
        canvas = synthetic_images(1, 760, 680, seed=seed + (i // per_sequence if shared_view0 else i))[0, 0]
        off1 = (16 + 4 * (i % per_sequence), 24 + 6 * (i % per_sequence)) if shared_view0 else (16, 24)
        views = {}
        for tag, (h, w), (y, x), up in (("view0", s0, (0, 0), 2.0), ("view1", s1, off1, 1.5)):
            g = canvas[y:y + h, x:x + w]
            rgb = torch.stack([g * 0.8, g, g * 0.9], 0).clamp(0, 1)
            rgb = ((rgb * 255).round() / 255).float()[None]  # what a decoded uint8 image gives
            ow, oh = int(w * up), int(h * up)
            views[tag] = {"image": rgb.to(device), "image_size": torch.tensor([[float(w), float(h)]], device=device),
                          "scales": torch.tensor([[w / ow, h / oh]], dtype=torch.float32, device=device),
                          "original_image_size": torch.tensor([[float(ow), float(oh)]], device=device)}
        items.append({"name": [f"synth{i // per_sequence}/{i % per_sequence + 2}.ppm"], "scene": [f"synth{i // per_sequence}"],
                      **views})
    return items


# (h, w) of decoded images whose short side 480 resize (ImagePreprocessor resize=480, side="short") gives
# HPATCHES_LIKE_SHAPES, in the same order
HPATCHES_LIKE_ORIGINALS = [(960, 1280), (960, 1226), (1280, 960), (1088, 720), (720, 984)]


def hpatches_like_host_images(n_pairs: int, seed: int = 4000, per_sequence: int = 5, pin: bool = True,
                              shared_view0: bool = False):
    """`n_pairs` RAW loader items as the HPatches dataset holds them right after decoding (datasets/hpatches.py:94-96:
    cv2.imread -> RGB uint8 [H,W,3]), at original sizes HPATCHES_LIKE_ORIGINALS, in (pinned) host memory: the input of
    image_preprocessor.HostImageFeeder.  Same sequence structure and image content as `hpatches_shaped_pairs` (the
    band-limited canvas is up-sampled bicubically to the original size and quantised to bytes).  shared_view0: the five pairs
    of a sequence carry the same view-0 image (one tensor), as in the real list; items then carry `scene`."""
    items = []
    ref_view = {}
    for i in range(n_pairs):
        j0 = (i // per_sequence) % len(HPATCHES_LIKE_SHAPES)
        j1 = (i * 2 + 1) % len(HPATCHES_LIKE_SHAPES)
        canvas = synthetic_images(1, 760, 680, seed=seed + (i // per_sequence if shared_view0 else i))[0, 0]
        off1 = (16 + 4 * (i % per_sequence), 24 + 6 * (i % per_sequence)) if shared_view0 else (16, 24)
        views = {}
        for tag, j, (y, x) in (("view0", j0, (0, 0)), ("view1", j1, off1)):
            if shared_view0 and tag == "view0" and i // per_sequence in ref_view:
                views[tag] = {"image": ref_view[i // per_sequence]}
                continue
            h, w = HPATCHES_LIKE_SHAPES[j]
            oh, ow = HPATCHES_LIKE_ORIGINALS[j]
            g = F.interpolate(canvas[None, None, y:y + h, x:x + w], size=(oh, ow), mode="bicubic", align_corners=False)[0, 0]
            rgb = torch.stack([g * 0.8, g, g * 0.9], -1).clamp(0, 1)
            u8 = (rgb * 255).round().to(torch.uint8).contiguous()
            views[tag] = {"image": u8.pin_memory() if pin else u8}
            if shared_view0 and tag == "view0":
                ref_view[i // per_sequence] = views[tag]["image"]
        items.append({"name": f"synth{i // per_sequence}/{i % per_sequence + 2}.ppm", "scene": f"synth{i // per_sequence}", **views})
    return items

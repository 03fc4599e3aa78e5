"""GPU counterpart of `gluefactory.utils.image.ImagePreprocessor` (reference gluefactory/utils/image.py:15-132): same
configuration keys, same returned dict (`image`, `scales`, `image_size`, `transform`, `original_image_size`, optional
square padding and `padding_mask`).  The resize itself -- kornia's antialiased bilinear resize in the reference -- runs
in `gfc_preprocess_resize`; it also accepts the decoded uint8 HxWxC image directly, fusing `numpy_image_to_torch`
(image.py:148-156).  Decoding files (cv2.imread) stays on the host and is not part of this module.

`HostImageFeeder` is the loader-side counterpart for the evaluation loop (datasets/hpatches.py:94-112 reads and
preprocesses both images of a pair on CPU worker processes, export_predictions.py:36-37 then copies the float images to
the device): decoded uint8 images in pinned host memory -> asynchronous host-to-device copies on a copy stream ->
`gfc_preprocess_resize` on the consumer's stream -> the loader item the export loop expects, tensors on the GPU.
"""
import collections.abc as collections
from collections import deque

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


class HostImageFeeder:
    """Iterable of loader items for `export_predictions` fed from DECODED uint8 images on the host.

    raw_items: sequence of dicts like the HPatches dataset's items before preprocessing: `{"name": str, "view0":
    {"image": uint8 [H,W,3] / [H,W] host tensor (RGB; pinned memory makes its copy asynchronous)}, "view1": {...},
    ...other keys passed through}`.  Per item and view the work of `HPatches._read_image` after decoding
    (numpy_image_to_torch + ImagePreprocessor, image.py:33-72,148-156) happens on the GPU: the bytes travel on a copy
    stream, `depth` items ahead of the consumer; the fused conversion + antialiased resize runs on the consumer's stream
    behind the copy's event; `scales` / `image_size` / `original_image_size` (host arithmetic) follow in one small
    pinned copy per item.  No host synchronisation anywhere: the export loop's kernels and the next items' copies overlap.
    Yields batch-1 items (`image` [1,C,h,w] float32, `scales` [1,2], `image_size` [1,2], `original_image_size` [1,2],
    `transform` [1,3,3] float64 on the host) with the keys and shapes the reference's DataLoader collates.  One
    difference, deliberate: the reference's `image_size` / `original_image_size` are integer arrays (image.py:52-58) that
    collate to int64 HOST tensors; here they are float32 DEVICE tensors holding the same integers (slices of the item's one
    small meta copy), which is what the extractor and matcher consume (lightglue.py:28-40 divides them) without a
    second copy or a host synchronisation.  Code that writes them into records gets floats (tests/test_preprocess.py).

    raw_items may be any iterable; `len()` and `shard()` index it when it is a sequence, otherwise `shard()` walks it
    and skips the other ranks' items (nothing of theirs is copied), and `len()` raises TypeError as the iterable's
    would.  Every iteration keeps its own ring and name table: two live iterators over one feeder do not interact."""

    def __init__(self, raw_items, conf, device="cuda", depth=64, bgr=False, view_key=None, keep=16):
        """view_key (optional): `view_key(raw_item, i) -> hashable or None`, the NAME of the image of view i; a name seen among
        the last `keep` named images is neither copied nor resized again -- the item gets the tensors of its first occurrence
        (an HPatches sequence names its image 1 as view 0 of all five of its pairs, datasets/hpatches.py:98-99)."""
        self.raw, self.pre, self.depth, self.bgr = raw_items, ImagePreprocessor(conf), max(1, int(depth)), bool(bgr)
        self.view_key, self.keep = view_key, max(1, int(keep))
        if self.pre.conf["square_pad"]:
            raise NotImplementedError("square_pad on the host-image path")
        if self.pre.conf["interpolation"] != "bilinear":
            raise NotImplementedError("only 'bilinear' is built on the GPU path")
        self.device = torch.device(device if device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        self.h2d_bytes = 0

    def __len__(self):
        return len(self.raw)

    def _stage(self, raw, copy_stream, slot, state):
        """Issue the copies of one item on the copy stream; returns what `_finish` needs.  state: this iteration's
        (name table, (pinned meta ring, its events))."""
        named, meta_ring = state
        views, meta, srcs = {}, [], []
        for i, tag in enumerate(("view0", "view1")):
            key = self.view_key(raw, i) if self.view_key is not None else None
            if key is not None and key in named:
                views[tag] = named[key]  # the first occurrence's record: filled by ITS _finish, which runs before ours
                meta += [0.0] * 6
                continue
            u8 = raw[tag]["image"]
            if u8.dtype != torch.uint8 or u8.ndim not in (2, 3):
                raise ValueError(f"{tag}: expected a decoded uint8 image [H,W,C] or [H,W], got {u8.dtype} {tuple(u8.shape)}")
            h, w = int(u8.shape[0]), int(u8.shape[1])
            size = (h, w) if self.pre.conf["resize"] is None else tuple(self.pre.get_new_image_size(h, w))
            # torch.Tensor([new_w / w, new_h / h]) (image.py:49): python floats rounded to fp32
            meta += [size[1] / w, size[0] / h, float(size[1]), float(size[0]), float(w), float(h)]
            srcs.append((tag, u8, size, (h, w), key))
            self.h2d_bytes += u8.numel()
        # the item's 12 numbers go through a slot of a pinned ring (allocated once); a slot comes round again after
        # `depth` + 1 items, when its copy has long completed (checked: the event is synchronised, which returns at once)
        ring, events = meta_ring
        if events[slot] is not None:
            events[slot].synchronize()
        ring[slot] = torch.tensor(meta, dtype=torch.float32)
        with torch.cuda.stream(copy_stream):
            for tag, u8, size, hw, key in srcs:
                rec = {"dev": u8.to(self.device, non_blocking=True), "size": size, "hw": hw, "out": None}
                views[tag] = rec
                if key is not None:
                    named[key] = rec
                    while len(named) > self.keep:
                        named.pop(next(iter(named)))  # oldest name first (insertion order)
            dev_meta = ring[slot].to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy_stream)
        events[slot] = done
        return raw, views, dev_meta, done

    def _finish(self, staged):
        raw, views, dev_meta, done = staged
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        dev_meta.record_stream(main)
        # strings as the DataLoader collates them: lists of one
        item = {k: ([v] if isinstance(v, str) else v) for k, v in raw.items() if k not in ("view0", "view1")}
        for j, tag in enumerate(("view0", "view1")):
            rec = views[tag]
            if rec["out"] is None:  # first (or only) occurrence of this image: convert + resize now
                dev, size, (h, w) = rec["dev"], rec["size"], rec["hw"]
                dev.record_stream(main)
                conf = self.pre.conf
                if conf["resize"] is not None:
                    img = resize(dev, size, conf["align_corners"], conf["antialias"], bgr=self.bgr)
                else:
                    img = resize(dev, size, None, False, bgr=self.bgr)  # conversion only
                m = dev_meta[6 * j: 6 * j + 6]
                sx, sy = size[1] / w, size[0] / h
                rec["out"] = {"image": img[None], "scales": m[0:2][None], "image_size": m[2:4][None],
                              "original_image_size": m[4:6][None],
                              "transform": torch.from_numpy(np.diag([np.float32(sx), np.float32(sy), 1.0]))[None]}
                rec["dev"] = None  # the bytes are not needed any more
            extra = {k: v for k, v in raw[tag].items() if k != "image"}
            item[tag] = {**extra, **rec["out"]}
        return item

    def __iter__(self):
        return self._iterate(self.raw)

    def shard(self, rank, world, group=1):
        """(index, item) of this rank's round-robin share (export_predictions' sharded mode, sharding.round_robin_shard:
        `group` consecutive items at a time): only this rank's images are copied and resized."""
        from .sharding import round_robin_shard
        rank, world, group = int(rank), int(world), max(1, int(group))
        if hasattr(self.raw, "__getitem__") and hasattr(self.raw, "__len__"):
            idx = round_robin_shard(len(self.raw), rank, world, group)
            return zip(idx, self._iterate(self.raw[i] for i in idx))
        # any other iterable (e.g. a DataLoader with batch_size=None): walk it, stage only this rank's items
        mine_idx, mine = [], []

        def own():
            for i, raw in enumerate(self.raw):
                if (i // group) % world == rank:
                    mine_idx.append(i)
                    yield raw

        def pairs():
            for n, item in enumerate(self._iterate(own())):
                yield mine_idx[n], item

        return pairs()

    def _iterate(self, raw_iterable):
        copy_stream = torch.cuda.Stream(self.device)
        ring = self.depth + 1
        state = ({}, (torch.empty((ring, 12), dtype=torch.float32).pin_memory(), [None] * ring))
        pending = deque()
        it = iter(raw_iterable)
        exhausted = False
        n_staged = 0
        while True:
            while not exhausted and len(pending) < self.depth:
                try:
                    pending.append(self._stage(next(it), copy_stream, n_staged % ring, state))
                    n_staged += 1
                except StopIteration:
                    exhausted = True
            if not pending:
                return
            yield self._finish(pending.popleft())

"""HPatches sequences read from a directory, feeding the GPU path -- counterpart of `gluefactory.datasets.hpatches`
(reference gluefactory/datasets/hpatches.py:23-35 `read_homography`, :38-77 sequence list, :94-112 items) and of the
file decode it relies on (gluefactory/utils/image.py:135-161: `cv2.imread` -> RGB uint8 -> float / 255).

HPatches ships binary PPM files (`<seq>/1.ppm` .. `6.ppm`) and plain-text 3x3 homographies (`<seq>/H_1_<q>`); the
files are decoded by `image_io.read_image` (a binary PPM is raw samples behind a text header: exact by construction, no
image library involved).  `HPatches` lists the pairs exactly as the reference's dataset does and yields RAW items --
decoded uint8 images in (pinned) host memory plus `H_0to1`, `scene`, `idx`, `is_illu`, `name` -- for
`image_preprocessor.HostImageFeeder`, which copies them to the GPU and does the reference's ImagePreprocessor work there.
No download: the reference fetches the archive when the directory is missing (hpatches.py:57,79-88); here a missing
directory is an error.
"""
import os
from pathlib import Path

import numpy as np
import torch

from .base_model import merge
from .image_io import image_size, read_image, read_ppm  # noqa: F401  (read_ppm re-exported: the files HPatches ships)
from .image_preprocessor import DEFAULT_CONF as PREPROCESSING_DEFAULTS
from .image_preprocessor import HostImageFeeder, ImagePreprocessor


def read_homography(path) -> np.ndarray:
    """Whitespace-separated 3x3 text matrix -> float64 array (hpatches.py:23-35: runs of spaces, trailing spaces and
    empty lines are tolerated)."""
    rows = []
    with open(path) as f:
        for line in f.readlines():
            elements = [e for e in line.replace("\n", "").split(" ") if e]
            if elements:
                rows.append(elements)
    return np.array(rows).astype(float)


class HPatches:
    """The reference dataset's pair list over a local `hpatches-sequences-release` directory, as a SEQUENCE of raw
    loader items (`len()`, indexing, iteration): item i = pair (`<seq>/1.ppm`, `<seq>/<q>.ppm`), q = 2..6."""

    default_conf = {
        "preprocessing": PREPROCESSING_DEFAULTS,
        "data_dir": "hpatches-sequences-release",  # absolute, or relative to $GFC_DATA_PATH / the current directory
        "subset": None,              # "i" (illumination) or "v" (viewpoint) sequences only
        "ignore_large_images": True,
        "grayscale": False,
        "pin_memory": True,          # decoded images in pinned host memory: their copies to the GPU are asynchronous
    }
    # hpatches.py:46-56 (spelling as in the reference: these are directory names)
    ignored_scenes = ("i_contruction", "i_crownnight", "i_dc", "i_pencils", "i_whitebuilding", "v_artisans",
                      "v_astronautis", "v_talent")

    def __init__(self, conf=None):
        self.conf = conf = merge(self.default_conf, dict(conf or {}))
        self.preprocessor = ImagePreprocessor(conf["preprocessing"])
        root = Path(conf["data_dir"])
        if not root.is_absolute():
            root = Path(os.environ.get("GFC_DATA_PATH", ".")) / root
        if not root.is_dir():
            raise FileNotFoundError(f"HPatches directory {root} not found (no download is attempted: the dataset must "
                                    "be on the machine)")
        self.root = root
        self.sequences = sorted(x.name for x in root.iterdir())
        if not self.sequences:
            raise ValueError("No image found!")
        self.items = []  # (seq, q_idx, is_illu)
        for seq in self.sequences:
            if conf["ignore_large_images"] and seq in self.ignored_scenes:
                continue
            if conf["subset"] is not None and conf["subset"] != seq[0]:
                continue
            for i in range(2, 7):
                self.items.append((seq, i, seq[0] == "i"))

    def __len__(self):
        return len(self.items)

    def _transform(self, h, w):
        """`T` of ImagePreprocessor.__call__ (image.py:49-50): diag of the fp32-rounded resize scales, as float64."""
        size = (h, w) if self.preprocessor.conf["resize"] is None else tuple(self.preprocessor.get_new_image_size(h, w))
        return np.diag([np.float32(size[1] / w), np.float32(size[0] / h), 1.0]).astype(np.float64)

    def _read(self, seq, idx):
        """Decoded uint8 image as a host tensor; with `pin_memory` the file is read straight into pinned memory (no
        pageable intermediate, no copy by torch's CPU thread pool)."""
        if not (self.conf["pin_memory"] and torch.cuda.is_available()):
            return torch.from_numpy(read_image(self.root / seq / f"{idx}.ppm", self.conf["grayscale"]))
        holder = []

        def alloc(nbytes):
            holder.append(torch.empty(nbytes, dtype=torch.uint8, pin_memory=True))
            return holder[0].numpy()

        arr = read_image(self.root / seq / f"{idx}.ppm", self.conf["grayscale"], alloc)
        return holder[0].view(arr.shape)

    def __getitem__(self, idx):
        """Raw item of pair `idx` (hpatches.py:98-112), already in the collated shapes a batch-1 DataLoader gives:
        `H_0to1` [1,3,3] float32 in the coordinates of the PREPROCESSED images (T1 . H . T0^-1), `idx` [1], `is_illu` [1],
        `scene` / `name` strings (the feeder wraps them in lists), view images as decoded uint8 [H,W,3] (or [H,W])."""
        seq, q_idx, is_illu = self.items[idx]
        img0, img1 = self._read(seq, 1), self._read(seq, q_idx)
        H = read_homography(self.root / seq / f"H_1_{q_idx}")
        H = self._transform(*img1.shape[:2]) @ H @ np.linalg.inv(self._transform(*img0.shape[:2]))
        return {"H_0to1": torch.from_numpy(H.astype(np.float32))[None], "scene": seq, "idx": torch.tensor([idx]),
                "is_illu": torch.tensor([is_illu]), "name": f"{seq}/{idx}.ppm", "view0": {"image": img0},
                "view1": {"image": img1}}

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def meta(self, idx):
        """Pair `idx` without its pixels (file headers only): what the evaluation needs beside the cached predictions
        (eval/hpatches.py:125-160 reads `H_0to1`, `name`, `scene`, `view0.image_size` and, through CacheLoader, the views'
        `scales`).  Un-batched, on the host: `H_0to1` [3,3] float32, `scales` / `image_size` [2] float32."""
        seq, q_idx, is_illu = self.items[idx]
        views, T = {}, []
        for tag, i in (("view0", 1), ("view1", q_idx)):
            h, w = image_size(self.root / seq / f"{i}.ppm")
            size = (h, w) if self.preprocessor.conf["resize"] is None else tuple(self.preprocessor.get_new_image_size(h, w))
            T.append(self._transform(h, w))
            views[tag] = {"scales": torch.tensor([size[1] / w, size[0] / h], dtype=torch.float32),
                          "image_size": torch.tensor([float(size[1]), float(size[0])]),
                          "original_image_size": torch.tensor([float(w), float(h)])}
        H = T[1] @ read_homography(self.root / seq / f"H_1_{q_idx}") @ np.linalg.inv(T[0])
        return {"H_0to1": torch.from_numpy(H.astype(np.float32)), "scene": seq, "idx": idx, "is_illu": is_illu,
                "name": f"{seq}/{idx}.ppm", **views}

    def feeder(self, device="cuda", depth=64, keep=4, num_workers=2):
        """The loader for `export_predictions`: items preprocessed on the GPU; the sequence's image 1 -- view 0 of all
        five of its pairs -- is copied and resized once (`view_key`, also what `export_predictions(view_key=...)` takes
        to extract it once).  num_workers > 0: the files are read `num_workers` at a time by reader THREADS, ahead of the
        consumer (a file read releases the interpreter lock and lands straight in pinned memory, so threads do what the
        reference's 16 loader processes do, eval/hpatches.py:47, without shipping 7 MB items between processes: measured
        140 pairs/s through a DataLoader with 8 worker processes, 1770 from one thread, `profiles/r06_hpatches_from_files.txt`);
        under torch.distributed every rank reads only its own share of the list.  depth: items copied ahead of the
        consumer -- at least twice the consumer's pair batch, or staging a batch waits for its own copies."""
        return HPatchesFeeder(self, device=device, depth=depth, keep=keep, num_workers=int(num_workers))

    def raw_loader(self, indices=None, num_workers=0, prefetch=48):
        """Raw items in list order (or of `indices`); num_workers > 0: read by that many threads, up to `prefetch` items
        ahead of the consumer, yielded in order."""
        idx = list(range(len(self)) if indices is None else indices)
        if num_workers <= 0:
            for i in idx:
                yield self[i]
            return
        import itertools
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(num_workers, thread_name_prefix="gfc-hpatches-read") as ex:
            it = iter(idx)
            pending = deque(ex.submit(self.__getitem__, i) for i in itertools.islice(it, max(1, int(prefetch))))
            while pending:
                item = pending.popleft().result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(ex.submit(self.__getitem__, nxt))
                yield item

    @staticmethod
    def view_key(raw, i):
        scene = raw["scene"][0] if isinstance(raw["scene"], (list, tuple)) else raw["scene"]
        return (scene, 1) if i == 0 else None


class HPatchesFeeder(HostImageFeeder):
    """`HostImageFeeder` over an `HPatches` list whose files are read ahead of the consumer by reader threads."""

    def __init__(self, dataset, device="cuda", depth=64, keep=4, num_workers=2):
        super().__init__(dataset, dataset.conf["preprocessing"], device=device, depth=depth, view_key=dataset.view_key,
                         keep=keep)
        self.dataset, self.num_workers = dataset, num_workers

    def __iter__(self):
        return self._iterate(self.dataset.raw_loader(None, self.num_workers))

    def shard(self, rank, world, group=1):
        from .sharding import round_robin_shard
        idx = list(round_robin_shard(len(self.dataset), int(rank), int(world), max(1, int(group))))
        return zip(idx, self._iterate(self.dataset.raw_loader(idx, self.num_workers)))

"""Image files -> decoded uint8 arrays -- counterpart of `read_image` / `load_image` of the reference
(gluefactory/utils/image.py:135-161: `cv2.imread`, BGR -> RGB flip or IMREAD_GRAYSCALE, then `numpy_image_to_torch`).

  * PPM / PGM (binary P6 / P5, plain P3 / P2 -- what HPatches ships): read here, no library.  A binary PPM is a short
    text header followed by the raw samples, so the decode is exact by construction: the bytes of the file ARE the image.
  * everything else (PNG: the reference's assets; JPEG: MegaDepth-style data) through Pillow when it is importable (it is
    in this image; OpenCV is not).  PNG is lossless, so its pixels are the encoder's whatever library decodes them; a JPEG
    may differ from OpenCV's decode by one grey level here and there (different IDCT / up-sampling): unpinned.

Unpinned against OpenCV (absent): the GREY read of a COLOUR file (`grayscale=True`; not the evaluation's default,
datasets/hpatches.py:43) uses the 14-bit fixed-point weights OpenCV documents for BGR -> GRAY; 16-bit PPMs are refused.
The float conversion (`/ 255`, HWC -> CHW) is not done here: `image_preprocessor` takes the uint8 image and fuses it into
the GPU resize; `load_image` gives the reference's float tensor for callers that want it on the host.
"""
from pathlib import Path

import numpy as np
import torch


def _header_tokens(buf: bytes, count: int):
    """The first `count` whitespace-separated header tokens of a netpbm file ('#' starts a comment that runs to the end
    of the line) and the offset of the byte after the single whitespace that ends the last one."""
    tokens, pos, n = [], 0, len(buf)
    while len(tokens) < count:
        while pos < n and (buf[pos:pos + 1].isspace() or buf[pos:pos + 1] == b"#"):
            if buf[pos:pos + 1] == b"#":
                while pos < n and buf[pos:pos + 1] not in (b"\n", b"\r"):
                    pos += 1
            else:
                pos += 1
        start = pos
        while pos < n and not buf[pos:pos + 1].isspace() and buf[pos:pos + 1] != b"#":
            pos += 1
        if start == pos:
            raise IOError("truncated netpbm header")
        tokens.append(buf[start:pos])
    return tokens, pos + 1  # exactly one whitespace byte separates the header from the raster


def read_ppm(path, grayscale: bool = False, alloc=None) -> np.ndarray:
    """Binary (P6 / P5) or plain (P3 / P2) PPM / PGM file -> uint8 array, [H,W,3] RGB or, with `grayscale`, [H,W] --
    what `read_image` returns (gluefactory/utils/image.py:135-146: cv2.imread + BGR->RGB flip, or IMREAD_GRAYSCALE).
    A grey file read as colour has its channel repeated three times, as cv2.IMREAD_COLOR does.
    alloc (optional): `alloc(nbytes) -> writable uint8 array` for the result (e.g. a view of pinned host memory); a binary
    file whose layout is the result's is then read STRAIGHT into it (`readinto`: one pass over the bytes, no copy, and the
    interpreter lock is released while the file is read)."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"No image at path {path}.")
    with open(path, "rb") as f:
        head = f.read(512)
        magic = head[:2]
        if magic not in (b"P6", b"P5", b"P3", b"P2"):
            raise IOError(f"Could not read image at {path}.")  # the reference's error for a file cv2 cannot decode
        (_, w, h, maxval), off = _header_tokens(head, 4)
        w, h, maxval = int(w), int(h), int(maxval)
        if maxval != 255:
            raise NotImplementedError(f"{path}: maxval {maxval}; only 8-bit files (maxval 255) are read")
        c = 3 if magic in (b"P6", b"P3") else 1
        direct = (magic == b"P6" and not grayscale) or (magic == b"P5" and grayscale)
        if direct:  # the raster IS the result
            out = np.empty(h * w * c, np.uint8) if alloc is None else alloc(h * w * c)
            f.seek(off)
            if f.readinto(memoryview(out)) != h * w * c:
                raise IOError(f"Could not read image at {path}.")
            return out.reshape((h, w, 3) if c == 3 else (h, w))
        f.seek(0)
        buf = f.read()
    if magic in (b"P6", b"P5"):
        if len(buf) - off < h * w * c:
            raise IOError(f"Could not read image at {path}.")
        img = np.frombuffer(buf, np.uint8, h * w * c, off).reshape(h, w, c)
    else:
        vals = np.array(buf[off - 1:].split()[: h * w * c], dtype=np.int64)
        if vals.size != h * w * c:
            raise IOError(f"Could not read image at {path}.")
        img = vals.astype(np.uint8).reshape(h, w, c)
    if grayscale:
        res = img[..., 0] if c == 1 else _grey(img)
    else:
        res = np.repeat(img, 3, axis=2) if c == 1 else img
    if alloc is None:
        return np.ascontiguousarray(res).copy() if not res.flags.writeable else np.ascontiguousarray(res)
    out = alloc(res.size).reshape(res.shape)
    out[...] = res
    return out


def image_size(path):
    """(height, width) of an image file from its header alone (PPM / PGM here, other formats through Pillow's lazy open)."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"No image at path {path}.")
    with open(path, "rb") as f:
        head = f.read(512)
    if head[:2] in (b"P6", b"P5", b"P3", b"P2"):
        (_, w, h), _ = _header_tokens(head, 3)
        return int(h), int(w)
    from PIL import Image
    with Image.open(path) as im:
        return im.height, im.width


def _grey(rgb: np.ndarray) -> np.ndarray:
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)  # OpenCV's 14-bit BGR2GRAY weights


def read_image(path, grayscale: bool = False, alloc=None) -> np.ndarray:
    """Image file -> uint8 [H,W,3] RGB, or [H,W] with `grayscale` (gluefactory/utils/image.py:135-146, same errors).
    alloc: see `read_ppm` (honoured for every format)."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"No image at path {path}.")
    with open(path, "rb") as f:
        magic = f.read(2)
    if magic in (b"P6", b"P5", b"P3", b"P2"):
        return read_ppm(path, grayscale, alloc)
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover -- Pillow is part of the image
        raise IOError(f"Could not read image at {path}.") from e
    try:
        with Image.open(path) as im:
            if im.mode in ("L", "1", "I;16", "I") and im.mode != "L":
                im = im.convert("L")
            grey_file = im.mode == "L"
            arr = np.asarray(im if grey_file else im.convert("RGB"), dtype=np.uint8)
    except Exception as e:  # noqa: BLE001 -- whatever the decoder raises: the reference's error
        raise IOError(f"Could not read image at {path}.") from e
    if grayscale:
        res = arr if grey_file else _grey(arr)
    else:
        res = np.repeat(arr[..., None], 3, axis=2) if grey_file else arr
    out = np.empty(res.shape, np.uint8) if alloc is None else alloc(res.size).reshape(res.shape)
    out[...] = res
    return out


def numpy_image_to_torch(image: np.ndarray) -> torch.Tensor:
    """uint8 HxWxC / HxW -> float32 CxHxW in [0, 1] (image.py:148-156)."""
    if image.ndim == 3:
        image = image.transpose((2, 0, 1))
    elif image.ndim == 2:
        image = image[None]
    else:
        raise ValueError(f"Not an image: {image.shape}")
    return torch.tensor(image / 255.0, dtype=torch.float)


def load_image(path, grayscale: bool = False) -> torch.Tensor:
    """image.py:159-161."""
    return numpy_image_to_torch(read_image(path, grayscale=grayscale))

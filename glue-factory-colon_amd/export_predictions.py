"""Export loop -- counterpart of gluefactory/utils/export_predictions.py:21-91 for this package.

`for data in loader: pred = model(data)`; key filtering (`keys` / `optional_keys`, ValueError on a missing
key), un-scaling of key points to the original image resolution (`keypoints{i} * 1/view{i}.scales`), one record
per pair named `data["name"][0]` holding every exported key without its batch dimension.  The reference writes
HDF5 groups through h5py; h5py is used here too when it is importable, otherwise the same records go into
one `.npz` archive with keys `"<name>/<key>"` (`load_predictions` reads both).
"""
from pathlib import Path

import numpy as np
import torch


def _to_device(data, device):
    if isinstance(data, torch.Tensor):
        return data.to(device, non_blocking=True)
    if isinstance(data, dict):
        return {k: _to_device(v, device) for k, v in data.items()}
    if isinstance(data, (list, tuple)) and data and isinstance(data[0], torch.Tensor):
        return type(data)(_to_device(v, device) for v in data)
    return data


@torch.no_grad()
def export_predictions(loader, model, output_file, as_half=False, keys="*", callback_fn=None, optional_keys=()):
    assert keys == "*" or isinstance(keys, (tuple, list))
    optional_keys = list(optional_keys)
    output_file = Path(output_file)
    output_file.parent.mkdir(exist_ok=True, parents=True)
    device = "cuda" if torch.cuda.is_available() else "cpu"
    model = model.to(device).eval()
    records = {}
    for data_ in loader:
        data = _to_device(data_, device)
        name = data.get("name", [None])[0]
        pred = model(data)
        if callback_fn is not None:
            pred = {**callback_fn(pred, data), **pred}
        if keys != "*":
            missing = set(keys) - set(pred.keys())
            if missing:
                raise ValueError(f"Missing key {missing}")
            pred = {k: v for k, v in pred.items() if k in list(keys) + optional_keys}
        assert len(pred) > 0
        for k in list(pred.keys()):  # back to the resolution of the original image
            if k.startswith("keypoints"):
                idx = k.replace("keypoints", "")
                scales = 1.0 / (data["scales"] if len(idx) == 0 else data[f"view{idx}"]["scales"])
                pred[k] = pred[k] * scales[None]
        rec = {k: v[0].cpu().numpy() for k, v in pred.items()}
        if as_half:
            rec = {k: (v.astype(np.float16) if v.dtype == np.float32 else v) for k, v in rec.items()}
        if name in records:
            continue  # like the reference: a duplicate group name is skipped
        records[name] = rec
    _write(output_file, records)
    return output_file


def _write(path: Path, records: dict):
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None and path.suffix in (".h5", ".hdf5"):
        with h5py.File(str(path), "w") as f:
            for name, rec in records.items():
                grp = f.create_group(name)
                for k, v in rec.items():
                    grp.create_dataset(k, data=v)
        return
    flat = {f"{name}/{k}": v for name, rec in records.items() for k, v in rec.items()}
    with open(path, "wb") as fh:
        np.savez(fh, **flat)


def load_predictions(path):
    """{name: {key: ndarray}} from either container."""
    path = Path(path)
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic == b"\x89HDF":
        import h5py

        with h5py.File(str(path), "r") as f:
            out = {}

            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    grp, key = name.rsplit("/", 1)
                    out.setdefault(grp, {})[key] = obj[()]

            f.visititems(visit)
            return out
    out = {}
    with np.load(path) as z:
        for full in z.files:
            name, key = full.rsplit("/", 1)
            out.setdefault(name, {})[key] = z[full]
    return out

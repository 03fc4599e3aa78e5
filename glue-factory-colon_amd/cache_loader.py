"""Counterpart of `gluefactory.models.cache_loader.CacheLoader` (reference gluefactory/models/cache_loader.py:63-171):
a model that, instead of computing, reads the records written by `export_predictions` for `data["name"]`, moves them
to the device of the batch and re-applies the view's `scales` to key points (the exporter divided them out,
export_predictions.py:73-79), so that cached features feed the pipeline (`allow_no_extract`, `view{i}.cache`).

Same configuration keys.  Differences: the container may also be this package's `.npz` archive (h5py is optional
here); `padding_fn` takes the function NAME ("pad_local_features") instead of an `eval`-ed expression; paths are used
as given unless `add_data_path` and a `GFC_DATA_PATH` environment variable are set (the reference prefixes its
settings.DATA_PATH).
"""
import os
import string
from pathlib import Path

import torch

from .base_model import BaseModel, conf_get
from .export_predictions import load_predictions


def _pad_to_length(x: torch.Tensor, length: int, dim: int, mode: str) -> torch.Tensor:
    """models/utils/misc.py:19-62 (`pad_to_length`): zeros, uniform noise, or per-column uniform in [min, max]."""
    n = x.shape[dim]
    if n >= length:
        return x
    shape = list(x.shape)
    shape[dim] = length - n
    if mode == "zeros":
        pad = torch.zeros(shape, dtype=x.dtype, device=x.device)
    elif mode == "random":
        pad = torch.rand(shape, dtype=x.dtype, device=x.device)
    elif mode == "random_c":
        lo, hi = (x.min(dim=dim, keepdim=True).values, x.max(dim=dim, keepdim=True).values) if n > 0 else (0.0, 1.0)
        pad = torch.rand(shape, dtype=x.dtype, device=x.device) * (hi - lo) + lo
    else:
        raise ValueError(mode)
    return torch.cat([x, pad], dim)


def pad_local_features(pred: dict, seq_l: int) -> dict:
    """cache_loader.py:17-45: bring one image's features to a fixed number of key points for batching."""
    pred["keypoints"] = _pad_to_length(pred["keypoints"], seq_l, -2, "random_c")
    for key, dim, mode in (("keypoint_scores", -1, "zeros"), ("descriptors", -2, "random"), ("scales", -1, "zeros"),
                           ("oris", -1, "zeros"), ("depth_keypoints", -1, "zeros"), ("valid_depth_keypoints", -1, "zeros")):
        if key in pred:
            pred[key] = _pad_to_length(pred[key], seq_l, dim, mode)
    return pred


_PADDING_FNS = {"pad_local_features": pad_local_features}
_DTYPES = {None: None, "float16": torch.float16, "float32": torch.float32, "float64": torch.float64}


class CacheLoader(BaseModel):
    default_conf = {
        "path": "???",  # can be a format string like exports/{scene}/
        "data_keys": None,  # load all keys
        "device": None,  # load to same device as data
        "trainable": False,
        "add_data_path": True,
        "collate": True,
        "scale": ["keypoints", "lines", "orig_lines"],
        "padding_fn": None,
        "padding_length": None,  # required for batching!
        "numeric_type": "float32",
    }
    required_data_keys = ["name"]

    def _init(self, conf):
        fn = conf_get(conf, "padding_fn")
        if fn is not None and fn not in _PADDING_FNS:
            raise NotImplementedError(f"padding_fn {fn!r}: known functions are {sorted(_PADDING_FNS)}")
        self.padding_fn = _PADDING_FNS.get(fn)
        self.numeric_dtype = _DTYPES[conf_get(conf, "numeric_type")]
        self._files = {}
        self.set_initialized()

    def _records(self, fpath):
        if fpath not in self._files:
            self._files[fpath] = load_predictions(fpath)
        return self._files[fpath]

    def _forward(self, data):
        conf = self.conf
        device = conf_get(conf, "device")
        if not device:
            devices = {v.device for v in data.values() if isinstance(v, torch.Tensor)}
            assert len(devices) <= 1
            device = devices.pop() if devices else "cpu"
        path_t = str(conf_get(conf, "path"))
        var_names = [x[1] for x in string.Formatter().parse(path_t) if x[1]]
        preds = []
        for i, name in enumerate(data["name"]):
            fpath = Path(path_t.format(**{k: data[k][i] for k in var_names}))
            root = os.environ.get("GFC_DATA_PATH")
            if conf_get(conf, "add_data_path") and root and not fpath.is_absolute():
                fpath = Path(root) / fpath
            rec = self._records(str(fpath))[name]
            keys = conf_get(conf, "data_keys")
            pred = {k: torch.from_numpy(rec[k]) for k in (keys if keys is not None else rec.keys())}
            if self.numeric_dtype is not None:
                pred = {k: (v.to(self.numeric_dtype) if torch.is_floating_point(v) else v) for k, v in pred.items()}
            pred = {k: v.to(device) for k, v in pred.items()}
            for k in list(pred):
                for pattern in conf_get(conf, "scale"):
                    if k.startswith(pattern):
                        idx = k.replace(pattern, "")
                        scales = data["scales"] if len(idx) == 0 else data[f"view{idx}"]["scales"]
                        pred[k] = pred[k] * scales[i].to(pred[k])
            if self.padding_fn is not None:
                pred = self.padding_fn(pred, conf_get(conf, "padding_length"))
            preds.append(pred)
        if conf_get(conf, "collate"):
            return {k: torch.stack([p[k] for p in preds], 0) for k in preds[0]}
        assert len(preds) == 1
        return preds[0]

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = CacheLoader

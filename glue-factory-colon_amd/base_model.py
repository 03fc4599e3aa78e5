"""Module contract of the reference's models, without the omegaconf dependency.

Mirrors gluefactory/models/base_model.py:25-157: class-level `default_conf` merged through
the MRO, user conf merged on top (unknown keys tolerated: `strict_conf = False`),
`required_data_keys` checked with `AssertionError("Missing key ...")`, `_init / _forward /
loss` hooks, `is_initialized / set_initialized`.

When the real `gluefactory` package is importable, the boundary modules subclass ITS
`BaseModel` so that `TwoViewPipeline.is_initialized()` (base_model.py:137-151, required by
eval/io.py:88-98) recurses into them; otherwise they subclass the look-alike below.
Configurations may be plain dicts or DictConfig objects; values are read through `conf_get`.
"""
import copy
from collections.abc import Mapping

from torch import nn


class Conf(dict):
    """dict with attribute access and recursive wrapping (read side of a DictConfig)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            dict.__setitem__(self, k, Conf(v) if isinstance(v, Mapping) and not isinstance(v, Conf) else v)

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"Missing key {k} in configuration") from None

    def __setattr__(self, k, v):
        raise TypeError("configuration is read-only")

    def __deepcopy__(self, memo):
        return Conf({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_plain(conf):
    """DictConfig / Conf / dict -> plain nested dict."""
    if conf is None:
        return {}
    if isinstance(conf, Mapping):
        return {k: to_plain(v) if isinstance(v, Mapping) else (list(v) if _is_listconf(v) else v)
                for k, v in conf.items()}
    try:  # omegaconf.DictConfig is a Mapping-like without inheriting from it in old versions
        from omegaconf import OmegaConf

        return OmegaConf.to_container(conf, resolve=True)
    except Exception:  # pragma: no cover
        return dict(conf)


def _is_listconf(v):
    return type(v).__name__ == "ListConfig"


def merge(base, update):
    out = copy.deepcopy(to_plain(base))
    for k, v in to_plain(update).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


class _LocalBaseModel(nn.Module):
    default_conf = {
        "name": None,
        "trainable": True,
        "freeze_batch_normalization": False,
        "timeit": False,
    }
    required_data_keys = []
    strict_conf = False
    are_weights_initialized = False

    def __init__(self, conf):
        super().__init__()
        total = {}
        for klass in reversed(type(self).__mro__):
            dc = klass.__dict__.get("default_conf")
            if isinstance(dc, Mapping):
                total = merge(total, dc)
        user = to_plain(conf)
        if self.strict_conf:
            unknown = set(user) - set(total)
            if unknown:
                raise KeyError(f"unknown configuration entries {sorted(unknown)}")
        self.conf = conf = Conf(merge(total, user))
        self.required_data_keys = copy.copy(self.required_data_keys)
        self._init(conf)
        if not conf.trainable:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, data):
        def check(expected, given):
            for key in expected:
                assert key in given, f"Missing key {key} in data"
                if isinstance(expected, dict):
                    check(expected[key], given[key])

        check(self.required_data_keys, data)
        return self._forward(data)

    def _init(self, conf):
        raise NotImplementedError

    def _forward(self, data):
        raise NotImplementedError

    def loss(self, pred, data):
        raise NotImplementedError

    def load_state_dict(self, *args, **kwargs):
        ret = super().load_state_dict(*args, **kwargs)
        self.set_initialized()
        return ret

    def is_initialized(self):
        ok = True
        for _, w in self.named_children():
            if isinstance(w, _LocalBaseModel):
                ok = ok and w.is_initialized()
            else:
                n_params = len(list(w.parameters()))
                ok = ok and (n_params == 0 or self.are_weights_initialized)
        return ok

    def set_initialized(self, to: bool = True):
        self.are_weights_initialized = to
        for w in self.children():
            if isinstance(w, _LocalBaseModel):
                w.set_initialized(to)


def _reference_base():
    try:
        from gluefactory.models.base_model import BaseModel as RefBase  # needs omegaconf

        return RefBase
    except Exception:
        return None


_Ref = _reference_base()
#: `BaseModel` is the reference's own class when `gluefactory` is importable (drop-in inside a
#: glue-factory checkout), else the dependency-free look-alike.
BaseModel = _Ref if _Ref is not None else _LocalBaseModel
USING_REFERENCE_BASE = _Ref is not None


def conf_get(conf, key, default=None):
    try:
        v = conf[key]
    except (KeyError, AttributeError, TypeError):
        return default
    except Exception:  # omegaconf raises its own error types on missing keys
        return default
    return v

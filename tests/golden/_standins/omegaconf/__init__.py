"""Minimal stand-in for the `omegaconf` package (absent in the build container).

Test infrastructure only: it lets `tests/golden/make_golden.py` import the
reference's Python modules from /root/reference to generate golden vectors.
It is never imported by the product path (glue_factory_colon_amd has its own
plain-dict config handling).  API surface = what the reference's hot-path
modules touch.
"""
import contextlib
import copy

__all__ = ["OmegaConf", "DictConfig", "MissingMandatoryValue", "read_write", "open_dict"]


class MissingMandatoryValue(Exception):
    pass


def _wrap(v):
    if isinstance(v, DictConfig):
        return v
    if isinstance(v, dict):
        return DictConfig(v)
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    return v


class DictConfig(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            dict.__setitem__(self, k, _wrap(v))

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        if isinstance(v, str) and v == "???":
            raise MissingMandatoryValue(k)
        return v

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def __setitem__(self, k, v):
        dict.__setitem__(self, k, _wrap(v))

    def __deepcopy__(self, memo):
        return DictConfig({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _merge(a, b):
    out = DictConfig(a)
    for k, v in b.items():
        if k in out and isinstance(out[k], dict) and isinstance(v, dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(_wrap(v))
    return out


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_plain(x) for x in v]
    return v


class OmegaConf:
    @staticmethod
    def create(d=None):
        return DictConfig(d or {})

    @staticmethod
    def merge(*confs):
        out = DictConfig()
        for c in confs:
            out = _merge(out, c if isinstance(c, dict) else {})
        return out

    @staticmethod
    def set_struct(conf, flag):
        return None

    @staticmethod
    def set_readonly(conf, flag):
        return None

    @staticmethod
    def to_container(conf, resolve=False):
        return _plain(conf)

    @staticmethod
    def resolve(conf):
        return None

    @staticmethod
    def from_cli(args=None):
        return DictConfig()

    @staticmethod
    def load(path):
        import yaml

        with open(path) as f:
            return DictConfig(yaml.safe_load(f) or {})


@contextlib.contextmanager
def read_write(conf):
    yield conf


@contextlib.contextmanager
def open_dict(conf):
    yield conf

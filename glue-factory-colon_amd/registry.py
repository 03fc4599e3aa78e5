"""`get_model(name)`: the reference's discovery rule (gluefactory/models/__init__.py:7-30) for
this package.  Names of the reference's own hot-path modules are mapped onto their MI355X
counterparts, so an unmodified glue-factory `model:` config block selects the HIP path."""
import importlib

_PKG = __name__.rsplit(".", 1)[0]

ALIASES = {
    "extractors.superpoint_open": "superpoint_open",
    "superpoint_open": "superpoint_open",
    "gluefactory.models.extractors.superpoint_open": "superpoint_open",
    "gluefactory_nonfree.superpoint": "superpoint",
    "superpoint": "superpoint",
    "matchers.lightglue": "lightglue",
    "lightglue": "lightglue",
    "gluefactory.models.matchers.lightglue": "lightglue",
    "matchers.lightglue_pretrained": "lightglue_pretrained",
    "lightglue_pretrained": "lightglue_pretrained",
    "matchers.nearest_neighbor_matcher": "nearest_neighbor_matcher",
    "nearest_neighbor_matcher": "nearest_neighbor_matcher",
    "extractors.disk_kornia": "disk_kornia",
    "disk_kornia": "disk_kornia",
    "gluefactory.models.extractors.disk_kornia": "disk_kornia",
    "two_view_pipeline": "two_view_pipeline",
    "cache_loader": "cache_loader",
}


def get_model(name: str):
    from .base_model import BaseModel

    candidates = []
    if name in ALIASES:
        candidates.append(f"{_PKG}.{ALIASES[name]}")
    if name.startswith(_PKG + ".") or name.startswith("glue_factory_colon_amd."):
        candidates.append(name)
    for path in candidates:
        try:
            mod = importlib.import_module(path)
        except ModuleNotFoundError:
            continue
        if hasattr(mod, "__main_model__"):
            return mod.__main_model__
        classes = [c for c in vars(mod).values()
                   if isinstance(c, type) and c.__module__ == mod.__name__ and issubclass(c, BaseModel)]
        if len(classes) == 1:
            return classes[0]
    raise RuntimeError(f"Model {name} not found: this package only provides {sorted(set(ALIASES.values()))} "
                       "(everything else in glue-factory is out of scope)")

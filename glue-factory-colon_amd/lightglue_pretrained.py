"""Boundary look-alike of `gluefactory.models.matchers.lightglue_pretrained`
(reference gluefactory/models/matchers/lightglue_pretrained.py:1-37), which wraps the
third-party `lightglue` pip package (unpinned git HEAD, pyproject.toml:35).  That package is the
same architecture as the in-tree matcher and ships the same checkpoints (lightglue.py:394-401
accepts them), so this wrapper runs the MI355X LightGlue with the wrapper's configuration keys:
`features` picks the input dimension (superpoint: 256, disk: 128) and `weights` must point to a
local checkpoint ("synthetic[:seed]" = name-seeded weights; nothing is downloaded).

    model.matcher.name = glue_factory_colon_amd.lightglue_pretrained
"""
from .base_model import BaseModel, conf_get
from .lightglue import LightGlue as _LightGlue

_FEATURE_DIM = {"superpoint": 256, "disk": 128, "aliked": 128, "sift": 128, "doghardnet": 128}


class LightGlue(BaseModel):
    default_conf = {"features": "superpoint", "depth_confidence": -1, "width_confidence": -1,
                    "filter_threshold": 0.1, "weights": None}
    required_data_keys = ["view0", "keypoints0", "descriptors0", "view1", "keypoints1", "descriptors1"]

    def _init(self, conf):
        feats = conf_get(conf, "features")
        if feats not in _FEATURE_DIM:
            raise ValueError(f"unknown features {feats!r}")
        self.net = _LightGlue({
            "input_dim": _FEATURE_DIM[feats],
            # the third-party package's feature table gives sift / doghardnet `add_scale_ori` (scales0/1, oris0/1
            # forwarded by the wrapper: lightglue_pretrained.py:24-33)
            "add_scale_ori": feats in ("sift", "doghardnet"),
            "depth_confidence": conf_get(conf, "depth_confidence"),
            "width_confidence": conf_get(conf, "width_confidence"),
            "filter_threshold": conf_get(conf, "filter_threshold"),
            "weights": conf_get(conf, "weights"),
        })
        if self.net.are_weights_initialized:
            self.set_initialized()

    def load_state_dict(self, state_dict, *args, **kwargs):
        """Accepts both the wrapper's `net.*` keys and a bare LightGlue checkpoint."""
        if not any(k.startswith("net.") for k in state_dict):
            ret = self.net.load_state_dict(state_dict, *args, **kwargs)
            self.set_initialized()
            return ret
        ret = super().load_state_dict(state_dict, *args, **kwargs)
        self.net._packed = None
        self.net.are_weights_initialized = True
        self.set_initialized()
        return ret

    def _forward(self, data):
        # lightglue_pretrained.py:23-33 repacks {view, keypoints, descriptors} per image; the
        # in-tree contract takes the same tensors under the pipeline's own keys
        return self.net(data)

    def forward_pairs(self, items):
        """Several batch-1 inputs of different sizes through one matcher pass (lightglue.LightGlue.forward_pairs)."""
        for data in items:
            for key in self.required_data_keys:
                assert key in data, f"Missing key {key} in data"
        return self.net.forward_pairs(items)

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = LightGlue

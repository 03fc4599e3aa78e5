"""SuperPoint (official arithmetic: conv -> ReLU, no BN, legacy descriptor sampling) on MI355X --
drop-in for `gluefactory_nonfree.superpoint` (behavioural spec:
gluefactory_nonfree/superpoint.py:155-385; that file carries a restrictive licence and was used
as a description of behaviour only).

Same configuration keys, data / prediction dictionaries and state-dict key names
(`conv1a` ... `convDb`), so `superpoint_v1.pth` loads unchanged.  Select it with
    model.extractor.name = glue_factory_colon_amd.superpoint
"""
from pathlib import Path

import torch
from torch import nn

from . import _native as nat
from . import weights as _weights
from ._superpoint_common import RaggedCounts, extract_views, joint_pair_data, SAMPLE_FIXED, SAMPLE_LEGACY, PackedSuperPoint, SuperPointRunner, run_extractor
from .base_model import BaseModel, conf_get

_LAYERS = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"]


class SuperPoint(BaseModel):
    default_conf = {
        "has_detector": True,
        "has_descriptor": True,
        "descriptor_dim": 256,
        "sparse_outputs": True,
        "dense_outputs": False,
        "nms_radius": 4,
        "refinement_radius": 0,
        "detection_threshold": 0.005,
        "max_num_keypoints": -1,
        "max_num_keypoints_val": None,
        "force_num_keypoints": False,
        "randomize_keypoints_training": False,
        "remove_borders": 4,
        "legacy_sampling": True,
        "filter_specular_keypoints": True,
        "weights": None,  # extension: local checkpoint path or "synthetic[:seed]" (the reference always downloads)
        "pad_random": "device",  # extension: "torch_cpu" = the reference's padding draws (see superpoint_open.py)
    }
    required_data_keys = ["image"]

    def _init(self, conf):
        if not (conf_get(conf, "has_detector") and conf_get(conf, "has_descriptor")):
            raise NotImplementedError("detector-only / descriptor-only variants are not built")
        if conf_get(conf, "descriptor_dim") != 256:
            raise NotImplementedError("descriptor_dim must be 256")
        c1, c2, c3, c4, c5 = 64, 64, 128, 128, 256
        self.conv1a = nn.Conv2d(1, c1, 3, 1, 1)
        self.conv1b = nn.Conv2d(c1, c1, 3, 1, 1)
        self.conv2a = nn.Conv2d(c1, c2, 3, 1, 1)
        self.conv2b = nn.Conv2d(c2, c2, 3, 1, 1)
        self.conv3a = nn.Conv2d(c2, c3, 3, 1, 1)
        self.conv3b = nn.Conv2d(c3, c3, 3, 1, 1)
        self.conv4a = nn.Conv2d(c3, c4, 3, 1, 1)
        self.conv4b = nn.Conv2d(c4, c4, 3, 1, 1)
        self.convPa = nn.Conv2d(c4, c5, 3, 1, 1)
        self.convPb = nn.Conv2d(c5, 65, 1, 1, 0)
        self.convDa = nn.Conv2d(c4, c5, 3, 1, 1)
        self.convDb = nn.Conv2d(c5, conf_get(conf, "descriptor_dim"), 1, 1, 0)
        self._packed = None
        self._runner = SuperPointRunner()
        w = conf_get(conf, "weights")
        if w is not None and Path(str(w)).exists():
            self.load_state_dict(torch.load(str(w), map_location="cpu"), strict=False)
        elif isinstance(w, str) and w.startswith("synthetic"):
            seed = int(w.split(":")[1]) if ":" in w else 0
            self.load_state_dict(_weights.superpoint_state_dict(seed), strict=False)
        elif w is not None:
            raise FileNotFoundError(f"weights file {w!r} not found (no download is attempted)")

    def load_state_dict(self, *args, **kwargs):
        ret = super().load_state_dict(*args, **kwargs)
        self._packed = None
        self.set_initialized()
        return ret

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        return super()._apply(fn, *args, **kwargs)

    def _pack(self, device):
        def cv(m):
            return m.weight, m.bias, None, None

        return PackedSuperPoint([cv(getattr(self, n)) for n in _LAYERS], cv(self.convPa), cv(self.convDa),
                                cv(self.convPb), cv(self.convDb), device,
                                conv_mode=conf_get(self.conf, "conv_arithmetic", None))

    def ensure_packed(self, device):
        """The device copies of the weights in the library's layouts, built on the CALLING thread's current stream if
        they do not exist yet (see superpoint_open.SuperPoint.ensure_packed)."""
        if not self.are_weights_initialized:
            raise RuntimeError("SuperPoint weights are not loaded (conf.weights or load_state_dict)")
        if self._packed is None or self._packed.device != device:
            self._packed = self._pack(device)
        return self._packed

    def _max_keypoints(self):
        max_kps = conf_get(self.conf, "max_num_keypoints")
        if not self.training and conf_get(self.conf, "max_num_keypoints_val") is not None:
            max_kps = conf_get(self.conf, "max_num_keypoints_val")
        return None if (max_kps is None or max_kps <= 0) else int(max_kps)

    def defers_counts(self):
        """True when a per-image call can leave the key-point counts on the device (run_extractor, defer_counts)."""
        return (bool(conf_get(self.conf, "sparse_outputs")) and self._max_keypoints() is not None
                and not conf_get(self.conf, "force_num_keypoints"))

    def _forward(self, data, per_image=False, defer_counts=False, runner=None):
        if not self.are_weights_initialized:
            raise RuntimeError("SuperPoint weights are not loaded (conf.weights or load_state_dict)")
        conf = self.conf
        if self.training and conf_get(conf, "randomize_keypoints_training"):
            raise NotImplementedError("training-time multinomial sampling is out of scope (inference path)")
        specular = "after_topk" if ("specular_mask" in data and conf_get(conf, "filter_specular_keypoints")) else None
        nat.require_cuda(data["image"], "data['image']")
        self.ensure_packed(data["image"].device)
        if not conf_get(conf, "sparse_outputs"):
            with torch.no_grad():
                image = data["image"].float().contiguous()
                heat, desc_raw = self._runner.dense(self._packed, image)
                dense = self._runner.l2norm_rows(desc_raw)
            return {"keypoint_scores": heat, "descriptors": dense.permute(0, 3, 1, 2)}
        k = self._max_keypoints()
        with torch.no_grad():
            return run_extractor(
                runner or self._runner, self._packed, data,
                nms_radius=conf_get(conf, "nms_radius"), remove_borders=conf_get(conf, "remove_borders"),
                detection_threshold=conf_get(conf, "detection_threshold"), max_num_keypoints=k,
                force_num_keypoints=conf_get(conf, "force_num_keypoints"),
                pad_random=conf_get(conf, "pad_random", "device"),
                sample_mode=SAMPLE_LEGACY if conf_get(conf, "legacy_sampling") else SAMPLE_FIXED,
                use_image_size_for_borders=True, dense_outputs=conf_get(conf, "dense_outputs"), specular=specular,
                refinement_radius=conf_get(conf, "refinement_radius", 0) or 0, per_image=per_image, defer_counts=defer_counts)

    def forward_pair(self, data0, data1):
        """Both views of an image pair through ONE extractor call when their images agree in shape (see
        superpoint_open.SuperPoint.forward_pair).  Returns (pred0, pred1), each exactly what `self(view)` returns."""
        for d in (data0, data1):  # what BaseModel.forward checks for a single view (base_model.py:101-113)
            for key in self.required_data_keys:
                assert key in d, f"Missing key {key} in data"
        joint = joint_pair_data(data0, data1) if conf_get(self.conf, "sparse_outputs") else None
        if joint is None:
            return self(data0), self(data1)
        b = data0["image"].shape[0]
        if b == 1:  # the two views may yield different numbers of key points
            preds = self._forward(joint, per_image=True)
            return preds[0], preds[1]
        try:
            pred = self._forward(joint)  # batched views: one count for all 2b images
        except RaggedCounts:
            # without padding the counts need only agree INSIDE a view (the reference runs one call per view)
            return self(data0), self(data1)
        return {k: v[:b] for k, v in pred.items()}, {k: v[b:] for k, v in pred.items()}

    def forward_views(self, views):
        """Single-image inputs of DIFFERENT image shapes -> their predictions, one extractor call per distinct shape
        (_superpoint_common.extract_views).  MI355X addition used by TwoViewPipeline.forward_pairs."""
        if not conf_get(self.conf, "sparse_outputs"):
            return [self(v) for v in views]
        return extract_views(self, views)

    def loss(self, pred, data):
        raise NotImplementedError

    def metrics(self, pred, data):
        raise NotImplementedError


__main_model__ = SuperPoint

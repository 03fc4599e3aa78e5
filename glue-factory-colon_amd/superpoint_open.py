"""SuperPoint (open re-implementation) extractor on MI355X -- drop-in for
`gluefactory.models.extractors.superpoint_open` (reference file
gluefactory/models/extractors/superpoint_open.py:80-235).

Same configuration keys, same input / output dictionary, same state-dict key names
(`backbone.{b}.{j}.conv|bn.*`, `detector.{0,1}.*`, `descriptor.{0,1}.*`), so a real
`superpoint_v6_from_tf.pth` loads unchanged.  The torch sub-modules below are parameter
containers only: they are never called; the forward pass runs in libgfc_amd.so.

Select it from a glue-factory config with
    model.extractor.name = glue_factory_colon_amd.superpoint_open
"""
from collections import OrderedDict
from pathlib import Path

import torch
from torch import nn

from . import _native as nat
from . import weights as _weights
from ._superpoint_common import RaggedCounts, extract_views, joint_pair_data, SAMPLE_OPEN, PackedSuperPoint, SuperPointRunner, fold_bn, run_extractor
from .base_model import BaseModel, conf_get


def _vgg_block(c_in, c_out, kernel_size, relu=True):
    """Parameter container with the reference's child names (conv / activation / bn)."""
    return nn.Sequential(OrderedDict([
        ("conv", nn.Conv2d(c_in, c_out, kernel_size, stride=1, padding=(kernel_size - 1) // 2)),
        ("activation", nn.ReLU(inplace=True) if relu else nn.Identity()),
        ("bn", nn.BatchNorm2d(c_out, eps=0.001)),
    ]))


class SuperPoint(BaseModel):
    default_conf = {
        "descriptor_dim": 256,
        "nms_radius": 4,
        "max_num_keypoints": None,
        "force_num_keypoints": False,
        "detection_threshold": 0.005,
        "remove_borders": 4,
        "channels": [64, 64, 128, 128, 256],
        "dense_outputs": None,
        "weights": None,  # local path of pretrained weights; "synthetic[:seed]" = name-seeded weights
        "filter_specular_keypoints": True,
        # MI355X addition, arithmetic of the 3x3 convolutions: None = $GFC_CONV_MODE or "winograd" (Winograd F(2x2,3x3)
        # on fp32 MFMA: same fp32 products / accumulation, 2.25x fewer of them); "fp32" = direct implicit GEMM on fp32
        # MFMA
        "conv_arithmetic": None,
        # MI355X addition, source of the random padding of `force_num_keypoints` (models/utils/misc.py:48-60): "device" =
        # one launch with the library's own generator, no host synchronisation; "torch_cpu" = the reference's draws from
        # torch's CPU generator, bit for bit under the same torch.manual_seed (host round trip; parity runs)
        "pad_random": "device",
    }
    required_data_keys = ["image"]

    def _init(self, conf):
        channels = list(conf_get(conf, "channels"))
        if channels != [64, 64, 128, 128, 256] or conf_get(conf, "descriptor_dim") != 256:
            raise NotImplementedError("the MI355X kernels are built for channels [64,64,128,128,256], 256-d")
        self.stride = 2 ** (len(channels) - 2)
        chans = [1, *channels[:-1]]
        backbone = []
        for i, c in enumerate(chans[1:], 1):
            layers = [_vgg_block(chans[i - 1], c, 3), _vgg_block(c, c, 3)]
            if i < len(chans) - 1:
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            backbone.append(nn.Sequential(*layers))
        self.backbone = nn.Sequential(*backbone)
        c = channels[-1]
        self.detector = nn.Sequential(_vgg_block(chans[-1], c, 3), _vgg_block(c, self.stride ** 2 + 1, 1, relu=False))
        self.descriptor = nn.Sequential(_vgg_block(chans[-1], c, 3),
                                        _vgg_block(c, conf_get(conf, "descriptor_dim"), 1, relu=False))
        self._packed = None
        self._runner = SuperPointRunner()
        w = conf_get(conf, "weights")
        if w is not None and Path(str(w)).exists():
            self.load_state_dict(torch.load(str(w), map_location="cpu"))
        elif isinstance(w, str) and w.startswith("synthetic"):
            seed = int(w.split(":")[1]) if ":" in w else 0
            self.load_state_dict(_weights.superpoint_open_state_dict(seed))
        elif w is not None:
            # the reference downloads checkpoint_url here (superpoint_open.py:120-123); no network on this path
            raise FileNotFoundError(f"weights file {w!r} not found (no download is attempted)")
        # w is None: stay un-initialised until load_state_dict() is called

    # -- weight cache ------------------------------------------------------------------
    def load_state_dict(self, *args, **kwargs):
        ret = super().load_state_dict(*args, **kwargs)
        self._packed = None
        self.set_initialized()
        return ret

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        return super()._apply(fn, *args, **kwargs)

    def _pack(self, device):
        def blk(m):
            a, b = fold_bn(m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var, m.bn.eps)
            return m.conv.weight, m.conv.bias, a, b

        layers = [blk(self.backbone[b][j]) for b in range(4) for j in range(2)]
        return PackedSuperPoint(layers, blk(self.detector[0]), blk(self.descriptor[0]), blk(self.detector[1]),
                                blk(self.descriptor[1]), device,
                                conv_mode=conf_get(self.conf, "conv_arithmetic"))

    def ensure_packed(self, device):
        """The device copies of the weights in the library's layouts, built on the CALLING thread's current stream if
        they do not exist yet.  Whoever hands this module to another stream (extract_views' lanes, export workers)
        calls this first on the stream the others wait on: packing is a sequence of kernels like any other."""
        if not self.are_weights_initialized:
            raise RuntimeError("SuperPoint weights are not loaded (conf.weights or load_state_dict)")
        if self._packed is None or self._packed.device != device:
            self._packed = self._pack(device)
        return self._packed

    def defers_counts(self):
        """True when a per-image call can leave the key-point counts on the device (run_extractor, defer_counts)."""
        k = conf_get(self.conf, "max_num_keypoints")
        return k is not None and not conf_get(self.conf, "force_num_keypoints")

    def _forward(self, data, per_image=False, defer_counts=False, runner=None):
        if not self.are_weights_initialized:
            raise RuntimeError("SuperPoint weights are not loaded (conf.weights or load_state_dict)")
        specular = "before_topk" if ("specular_mask" in data and conf_get(self.conf, "filter_specular_keypoints")) else None
        nat.require_cuda(data["image"], "data['image']")
        self.ensure_packed(data["image"].device)
        with torch.no_grad():
            return run_extractor(
                runner or self._runner, self._packed, data,
                nms_radius=conf_get(self.conf, "nms_radius"),
                remove_borders=conf_get(self.conf, "remove_borders"),
                detection_threshold=conf_get(self.conf, "detection_threshold"),
                max_num_keypoints=conf_get(self.conf, "max_num_keypoints"),
                force_num_keypoints=conf_get(self.conf, "force_num_keypoints"),
                pad_random=conf_get(self.conf, "pad_random", "device"),
                sample_mode=SAMPLE_OPEN, use_image_size_for_borders=False,
                dense_outputs=conf_get(self.conf, "dense_outputs"), specular=specular, per_image=per_image, defer_counts=defer_counts)

    def forward_pair(self, data0, data1):
        """Both views of an image pair through ONE extractor call when their images agree in shape (MI355X addition
        used by TwoViewPipeline: at batch 1 the layers after the stem fill a fraction of the chip, two images
        fill twice as much; the two descriptor arrays also come out adjacent in memory, which the matcher reads
        without a copy).  Returns (pred0, pred1), each exactly what `self(view)` returns."""
        for d in (data0, data1):  # what BaseModel.forward checks for a single view (base_model.py:101-113)
            for key in self.required_data_keys:
                assert key in d, f"Missing key {key} in data"
        joint = joint_pair_data(data0, data1)
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
        return extract_views(self, views)

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = SuperPoint

"""DISK extractor on MI355X -- boundary look-alike of `gluefactory.models.extractors.disk_kornia`
(reference gluefactory/models/extractors/disk_kornia.py:10-140): same configuration keys, data / prediction
dictionaries and error behaviour.

The reference delegates ALL arithmetic to the third-party `kornia` package (`kornia.feature.DISK`, unpinned
>= 0.6.12): the U-Net, its pretrained weights ("depth": a download) and the detection functions.  kornia does not
exist offline; both halves are restated from kornia's published source and run as HIP kernels -- PARITY UNPINNED for
both (no reference run, no reference-held fixture), checked against oracle/disk_unet.py and oracle/disk.py:

    pad to a multiple of 16 (disk_kornia.py:31-35) -> network (disk_unet.py, csrc/disk_unet.hip: thin U-Net of 5x5
    convolutions on fp32 MFMA) -> crop -> window-5 NMS + cutoff -> top-n by the (n+1)-th score -> descriptors at the
    integer pixel, L2-normalised -> specular filter (offset 0.5) -> pad_and_stack -> +0.5   (disk_kornia.py:42-47,84-137;
    csrc/disk_detect.hip)

Weights: `self.model` carries kornia's parameter names (`model.unet.path_down...`), so a kornia DISK state dict -- or
the file kornia downloads for "depth", given as a local path -- loads with `load_state_dict`; `weights: "synthetic[:seed]"`
= name-seeded weights (tests, offline benchmarks).  "depth" itself needs a download: the module then stays
un-initialised (forward raises) until `load_state_dict` is called.  Nothing third-party is picked up implicitly: an
importable kornia is NOT used by itself.  A caller who wants another network passes it EXPLICITLY as
`DISK(conf, dense_fn=...)` / `set_dense_fn` with `dense_fn(images [b,3,H,W]) -> (heat-maps [b,1,H,W], descriptors
[b,D,H,W])` on the device.

    model.extractor.name = glue_factory_colon_amd.disk_kornia
"""
import time
from pathlib import Path

import torch

from . import _native as nat
from . import weights as _weights
from ._superpoint_common import pad_keypoints_native, specular_mask_bytes
from .base_model import BaseModel, conf_get
from .disk_unet import DiskUnet


class DISK(BaseModel):
    default_conf = {
        "weights": "depth",
        "dense_outputs": False,
        "max_num_keypoints": None,
        "desc_dim": 128,
        "nms_window_size": 5,
        "detection_threshold": 0.0,
        "force_num_keypoints": False,
        "pad_if_not_divisible": True,
        "chunk": 4,  # for reduced VRAM in training
        "filter_specular_keypoints": True,
    }
    required_data_keys = ["image"]

    def __init__(self, conf, dense_fn=None):
        object.__setattr__(self, "_dense_fn", dense_fn)  # a plain attribute (set before nn.Module.__init__ runs)
        super().__init__(conf)

    def _init(self, conf):
        self._ws = nat.Workspace()
        if self._dense_fn is not None:  # an explicitly supplied network replaces the native one
            self.set_initialized()
            return
        self.model = DiskUnet(int(conf_get(conf, "desc_dim")))
        w = conf_get(conf, "weights")
        if isinstance(w, str) and w.startswith("synthetic"):
            seed = int(w.split(":")[1]) if ":" in w else 0
            self.model.load_state_dict(_weights.disk_state_dict(seed, int(conf_get(conf, "desc_dim"))))
            self.set_initialized()
        elif w is not None and Path(str(w)).exists():
            ckpt = torch.load(str(w), map_location="cpu")
            self.model.load_state_dict(ckpt.get("extractor", ckpt))  # kornia's files keep the network under "extractor"
            self.set_initialized()
        # anything else ("depth", "epipolar": kornia downloads them, disk_kornia.py:25): no network here -- the module
        # stays un-initialised until load_state_dict() supplies the parameters

    def is_initialized(self):
        return bool(self.are_weights_initialized)

    def set_dense_fn(self, dense_fn):
        object.__setattr__(self, "_dense_fn", dense_fn)
        self.set_initialized()

    def load_state_dict(self, *args, **kwargs):
        ret = super().load_state_dict(*args, **kwargs)
        self.set_initialized()
        return ret

    # ---- network boundary: pad to /16, run, crop (disk_kornia.py:29-40) ----
    def _dense(self, images):
        """-> heat-map [n,h,w] contiguous, dense descriptors contiguous, their layout ("nhwc" native / "nchw" dense_fn)."""
        h, w = images.shape[2:]
        if conf_get(self.conf, "pad_if_not_divisible"):
            pd_h = 16 - h % 16 if h % 16 > 0 else 0
            pd_w = 16 - w % 16 if w % 16 > 0 else 0
            if pd_h or pd_w:
                images = torch.nn.functional.pad(images, (0, pd_w, 0, pd_h), value=0.0)  # plumbing: a zero-filled copy
        if self._dense_fn is not None:
            heat, desc = self._dense_fn(images)
            return (heat[..., :h, :w].reshape(heat.shape[0], h, w).contiguous().float(),
                    desc[..., :h, :w].contiguous().float(), "nchw")
        heat, desc = self.model.dense_nhwc(images)
        return heat[:, :h, :w].contiguous(), desc[:, :h, :w].contiguous(), "nhwc"

    def _forward(self, data):
        if not self.are_weights_initialized:
            raise RuntimeError("DISK: the network has no weights (conf.weights = 'synthetic' or a local file, "
                               "load_state_dict with kornia's DISK parameters, or an explicit dense_fn); nothing is downloaded")
        conf, lib = self.conf, nat.lib()
        image = data["image"]
        nat.require_cuda(image, "data['image']")
        dev, b = image.device, image.shape[0]
        k = conf_get(conf, "max_num_keypoints")
        window, cutoff = int(conf_get(conf, "nms_window_size")), float(conf_get(conf, "detection_threshold"))
        chunk = int(conf_get(conf, "chunk"))
        h, w = image.shape[2:]
        cap = int(k) if k is not None else h * w
        kpts = torch.empty((b, cap, 2), device=dev, dtype=torch.float32)
        ksc = torch.empty((b, cap), device=dev, dtype=torch.float32)
        counts = torch.empty((b,), device=dev, dtype=torch.int32)
        dense_all = []
        st = nat.stream_ptr(dev)
        core_ms = 0.0
        with torch.no_grad():
            for i in range(0, b, chunk):  # disk_kornia.py:62-83
                start = time.perf_counter()
                heat, dense, layout = self._dense(image[i:i + chunk].float())
                n_i = heat.shape[0]
                ws = self._ws.get(lib.gfc_disk_select_workspace_bytes(n_i, h, w), dev)
                nat.check(lib.gfc_disk_nms_select(nat.ptr(heat), n_i, h, w, window, cutoff, -1 if k is None else int(k),
                                                  cap, nat.ptr(kpts[i:i + n_i]), nat.ptr(ksc[i:i + n_i]),
                                                  nat.ptr(counts[i:i + n_i]), nat.ptr(ws), ws.numel(), st),
                          "gfc_disk_nms_select")
                core_ms += (time.perf_counter() - start) * 1e3
                dense_all.append(dense)
            if conf_get(conf, "filter_specular_keypoints") and "specular_mask" in data:
                # disk_kornia.py:84-107: filter(k + 0.5, offset 0.5) on integer pixels == the four-corner test at k
                smask, swh = specular_mask_bytes(data, b, dev)
                nat.check(lib.gfc_sp_filter_keypoints(nat.ptr(kpts), nat.ptr(ksc), nat.ptr(counts), b, cap, nat.ptr(smask),
                                                      smask.shape[-2], smask.shape[-1], nat.ptr(swh), 0.0, st),
                          "gfc_sp_filter_keypoints")
            d = int(dense_all[0].shape[1] if layout == "nchw" else dense_all[0].shape[3])
            gather = lib.gfc_disk_gather_descriptors if layout == "nchw" else lib.gfc_disk_gather_descriptors_nhwc
            force = conf_get(conf, "force_num_keypoints")
            if force:
                if k is None:
                    raise ValueError("force_num_keypoints needs max_num_keypoints")
                n_out = int(k)
            else:
                n = counts.tolist()  # host sync, as torch.stack of ragged lists in the reference
                if len(set(n)) != 1:
                    raise RuntimeError(f"images of one batch yield different numbers of keypoints {n}: "
                                       "use force_num_keypoints=True or batch size 1")
                n_out = n[0]
            # descriptors only for the slots that are returned: with max_num_keypoints = None the selection arrays have
            # one slot per pixel (cap = h*w), but n_out is a few thousand -- compact the key points first, then gather
            # into a [b, n_out, d] array (row capacity of both = n_out)
            if n_out < cap:
                kpts, ksc = kpts[:, :n_out].contiguous(), ksc[:, :n_out].contiguous()
            desc = torch.empty((b, n_out, d), device=dev, dtype=torch.float32)
            if n_out > 0:
                for j, i in enumerate(range(0, b, chunk)):  # slots >= count: zeros (pad_and_stack "zeros")
                    dn = dense_all[j]
                    nat.check(gather(nat.ptr(dn), dn.shape[0], d, h, w, nat.ptr(kpts[i:i + dn.shape[0]]),
                                     nat.ptr(counts[i:i + dn.shape[0]]), n_out, nat.ptr(desc[i:i + dn.shape[0]]), st),
                              "gfc_disk_gather_descriptors")
            if force:
                kpts, ksc = pad_keypoints_native(kpts, ksc, counts, n_out, 0, data, image)  # disk_kornia.py:109-124
        pred = {
            "keypoints": kpts.contiguous().to(image) + 0.5,
            "keypoint_scores": ksc.contiguous().to(image),
            "descriptors": desc.contiguous().to(image),
            "extractor_core_time_ms": image.new_full((b,), core_ms / b),
        }
        if conf_get(conf, "dense_outputs"):
            dense = torch.cat(dense_all, 0)
            pred["dense_descriptors"] = dense if layout == "nchw" else dense.permute(0, 3, 1, 2)  # [b,D,h,w] either way
        return pred

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = DISK

"""DISK extractor on MI355X -- boundary look-alike of `gluefactory.models.extractors.disk_kornia`
(reference gluefactory/models/extractors/disk_kornia.py:10-140): same configuration keys, data / prediction
dictionaries and error behaviour.

The reference delegates ALL arithmetic to the third-party `kornia` package (`kornia.feature.DISK`, unpinned
>= 0.6.12): the U-Net, its pretrained weights ("depth": a download) and the detection functions.  Neither kornia nor
the weights exist offline, so the network cannot be restated or pinned here.  What this module builds natively is
everything BEHIND the network, as HIP kernels (csrc/disk_detect.hip), restated from kornia's published source:

    pad to a multiple of 16 (disk_kornia.py:31-35) -> network -> crop -> window-5 NMS + cutoff -> top-n by the
    (n+1)-th score -> descriptors at the integer pixel, L2-normalised -> specular filter (offset 0.5) ->
    pad_and_stack -> +0.5                                                    (disk_kornia.py:42-47,84-137)

The network is supplied EXPLICITLY as `dense_fn(images [b,3,H,W]) -> (heat-maps [b,1,H,W], descriptors [b,D,H,W])`
on the device, passed as `DISK(conf, dense_fn=...)` or `set_dense_fn` (e.g. kornia's
`DISK.from_pretrained("depth").heatmap_and_dense_descriptors` where that package and its weights exist).  Nothing is
picked up implicitly: an importable kornia is NOT used by itself (no silent third-party eager network on the product
path), and without a dense_fn `forward` raises -- there is no substitute network.  NETWORK PARITY UNPINNED; the
post-network stages are tested against oracle/disk.py.

    model.extractor.name = glue_factory_colon_amd.disk_kornia
"""
import time

import torch

from . import _native as nat
from ._superpoint_common import pad_keypoints_native, specular_mask_bytes
from .base_model import BaseModel, conf_get


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
        if self._dense_fn is None:
            # No network is built into this package (kornia's U-Net source and weights are absent offline, a25 in
            # DESIGN.md), and kornia's PyTorch network is deliberately NOT picked up even when it is importable: the
            # product path never runs a third-party eager network silently.  A caller who wants that passes it
            # explicitly: DISK(conf, dense_fn=kornia.feature.DISK.from_pretrained("depth").heatmap_and_dense_descriptors).
            return  # not initialised: forward raises until a dense_fn is supplied
        self.set_initialized()

    def is_initialized(self):
        return self._dense_fn is not None and bool(self.are_weights_initialized)

    def set_dense_fn(self, dense_fn):
        object.__setattr__(self, "_dense_fn", dense_fn)
        self.set_initialized()

    def load_state_dict(self, *args, **kwargs):
        ret = super().load_state_dict(*args, **kwargs)
        if self._dense_fn is not None:
            self.set_initialized()
        return ret

    # ---- network boundary: pad to /16, run, crop (disk_kornia.py:29-40) ----
    def _dense(self, images):
        h, w = images.shape[2:]
        if conf_get(self.conf, "pad_if_not_divisible"):
            pd_h = 16 - h % 16 if h % 16 > 0 else 0
            pd_w = 16 - w % 16 if w % 16 > 0 else 0
            images = torch.nn.functional.pad(images, (0, pd_w, 0, pd_h), value=0.0)  # plumbing: a zero-filled copy
        heat, desc = self._dense_fn(images)
        return heat[..., :h, :w], desc[..., :h, :w]

    def _forward(self, data):
        if self._dense_fn is None:
            raise RuntimeError("DISK: no network available (kornia is not installed and no dense_fn was supplied); "
                               "the MI355X build provides the stages behind the network only")
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
                heat, dense = self._dense(image[i:i + chunk].float())
                heat = heat.reshape(heat.shape[0], h, w).contiguous().float()
                dense = dense.contiguous().float()
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
            d = int(dense_all[0].shape[1])
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
                    nat.check(lib.gfc_disk_gather_descriptors(nat.ptr(dn), dn.shape[0], d, h, w,
                                                              nat.ptr(kpts[i:i + dn.shape[0]]),
                                                              nat.ptr(counts[i:i + dn.shape[0]]), n_out,
                                                              nat.ptr(desc[i:i + dn.shape[0]]), st),
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
            pred["dense_descriptors"] = torch.cat(dense_all, 0)
        return pred

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = DISK

"""Host side shared by the two SuperPoint boundary modules: weight packing for the C ABI and the
launch sequence dense -> NMS -> select -> (pad) -> sample.  Tensor plumbing only; all arithmetic
happens in libgfc_amd.so.
"""
import ctypes
import os
import time

import torch

from . import _native as nat

SAMPLE_OPEN, SAMPLE_LEGACY, SAMPLE_FIXED = 0, 1, 2


class RaggedCounts(RuntimeError):
    """The images of one batched call kept different numbers of key points (no force_num_keypoints)."""


def fold_bn(weight, bias, mean, var, eps=1e-3):
    """Eval-mode BatchNorm as y = x*alpha + beta, with the same fp32 operation order as
    torch's CPU kernel (invstd = 1/sqrt(var+eps); alpha = invstd*weight; beta = bias - mean*alpha)."""
    invstd = 1.0 / torch.sqrt(var.float() + eps)
    alpha = invstd * weight.float()
    beta = bias.float() - mean.float() * alpha
    return alpha, beta


class PackedSuperPoint:
    """Device-resident weights in the layouts of include/gfc_amd.h (gfc_sp_params)."""

    def __init__(self, layers, head_p, head_d, pb, db, device, conv_mode=None):
        """layers: 8 tuples (w_oihw, bias, scale|None, shift|None) for conv1a..conv4b;
        head_p / head_d: 3x3 128->256 of detector / descriptor (same tuple form);
        pb / db: 1x1 heads (w [C,256,1,1], bias, scale|None, shift|None)."""
        lib = nat.lib()
        st = nat.stream_ptr(device)
        self.keep = []  # tensors referenced by raw pointers in the struct
        self.params = nat.SpParams()

        def dev(t):
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            self.keep.append(t)
            return t

        def pack3x3(w):
            w = dev(w)
            cout, cin = w.shape[0], w.shape[1]
            out = torch.empty((9, cout, cin), device=device, dtype=torch.float32)
            nat.check(lib.gfc_pack_conv3x3(nat.ptr(w), nat.ptr(out), cout, cin, st), "gfc_pack_conv3x3")
            self.keep.append(out)
            return out

        def opt(t):
            return None if t is None else dev(t)

        # 3x3 convolutions: "winograd" (default; F(2x2,3x3) on the fp32 matrix pipe, csrc/conv_wino.hip: fp32 products
        # and accumulation, 2.25x fewer of them) or "fp32" (direct implicit GEMM on fp32 MFMA, csrc/conv.hip).
        # $GFC_CONV_MODE overrides the default for modules that do not set conf.conv_arithmetic.  (The round-1
        # "split" arithmetic -- bf16x3-split MFMA products -- was retired from the library in round 4.)
        mode = conv_mode if conv_mode is not None else os.environ.get("GFC_CONV_MODE", "winograd")
        if mode not in ("fp32", "winograd"):
            raise ValueError(f"conv_mode {mode!r}: 'fp32' or 'winograd'")
        self.params.conv_mode = {"fp32": 0, "winograd": 2}[mode]

        def pack_wino(w):
            """Filters transformed for Winograd F(2x2,3x3) (G g G^T in float64, done by the library) in fragment order."""
            w = w.detach().to(device=device, dtype=torch.float32).contiguous()
            out = torch.empty((16 * w.shape[0] * w.shape[1],), device=device, dtype=torch.float32)
            nat.check(lib.gfc_pack_conv3x3_wino(nat.ptr(w), nat.ptr(out), w.shape[0], w.shape[1], st),
                      "gfc_pack_conv3x3_wino")
            self.keep.extend([w, out])
            return out

        for i, (w, b, sc, sh) in enumerate(layers):
            self.params.w[i] = pack3x3(w).data_ptr()
            if mode == "winograd" and i >= 1:
                self.params.w_wino[i] = pack_wino(w).data_ptr()
            if mode == "winograd" and i == 1 and tuple(w.shape[:2]) == (64, 64):
                # the stem's conv1b additionally as F(4x4,3x3) (csrc/conv_wino43.hip; $GFC_STEM_F43=0 selects F(2x2,3x3))
                w32 = w.detach().to(device=device, dtype=torch.float32).contiguous()
                w43 = torch.empty((36 * 64 * 64,), device=device, dtype=torch.float32)
                nat.check(lib.gfc_pack_conv3x3_wino43(nat.ptr(w32), nat.ptr(w43), 64, 64, st), "gfc_pack_conv3x3_wino43")
                self.keep.extend([w32, w43])
                self.params.w_stem_wino43 = w43.data_ptr()
            self.params.bias[i] = dev(b).data_ptr()
            sc, sh = opt(sc), opt(sh)
            self.params.scale[i] = sc.data_ptr() if sc is not None else None
            self.params.shift[i] = sh.data_ptr() if sh is not None else None
        # merged 3x3 heads
        wh = pack3x3(torch.cat([head_p[0], head_d[0]], 0))
        if mode == "winograd":
            self.params.wh_wino = pack_wino(torch.cat([head_p[0], head_d[0]], 0)).data_ptr()
        bh = dev(torch.cat([head_p[1], head_d[1]], 0))
        self.params.wh, self.params.bias_h = wh.data_ptr(), bh.data_ptr()
        if head_p[2] is not None:
            sch = dev(torch.cat([head_p[2], head_d[2]], 0))
            shh = dev(torch.cat([head_p[3], head_d[3]], 0))
            self.params.scale_h, self.params.shift_h = sch.data_ptr(), shh.data_ptr()
        else:
            self.params.scale_h = self.params.shift_h = None
        for tag, (w, b, sc, sh) in (("p", pb), ("d", db)):
            w = dev(w.reshape(w.shape[0], -1))
            setattr(self.params, "w" + tag, w.data_ptr())
            setattr(self.params, "bias_" + tag, dev(b).data_ptr())
            sc, sh = opt(sc), opt(sh)
            setattr(self.params, "scale_" + tag, sc.data_ptr() if sc is not None else None)
            setattr(self.params, "shift_" + tag, sh.data_ptr() if sh is not None else None)
        self.params.desc_dim = int(db[0].shape[0])
        self.desc_dim = self.params.desc_dim
        self.device = device


def pad_keypoints_native(kpts, scores, counts, k, low, data, image):
    """`pad_random_c` as ONE launch (gfc_sp_pad_keypoints), in place on the selection kernel's [B,cap] outputs; returns
    the [B,k] views."""
    lib = nat.lib()
    b, cap, _ = kpts.shape
    sizes = None
    if "image_size" in data:
        sizes = data["image_size"].to(device=kpts.device, dtype=torch.float32).contiguous()
    # seed = (seed, offset) of torch's device generator, whose offset is advanced like a consumer of randomness would:
    # reproducible under torch.manual_seed, no kernel launch, no synchronisation
    gen = torch.cuda.default_generators[kpts.device.index if kpts.device.index is not None else torch.cuda.current_device()]
    offset = gen.get_offset()
    gen.set_offset(offset + 4)
    seed = (gen.initial_seed() * 0x9E3779B1 + offset * 0x85EBCA77 + 1) & 0xFFFFFFFF
    nat.check(lib.gfc_sp_pad_keypoints(nat.ptr(kpts), nat.ptr(scores), nat.ptr(counts), b, cap, int(k), float(low),
                                       nat.ptr(sizes), 0 if sizes is None else sizes.numel(),
                                       float(min(image.shape[-2:])), seed, nat.stream_ptr(kpts.device)),
              "gfc_sp_pad_keypoints")
    if cap == k:
        return kpts, scores
    return kpts[:, :k].contiguous(), scores[:, :k].contiguous()


def pad_keypoints_torch_cpu(kpts, scores, counts, k, low, data, image):
    """`pad_and_stack(..., mode="random_c")` with THE REFERENCE'S random numbers (models/utils/misc.py:48-60,103-113):
    the reference pads on the device its tensors live on -- the CPU in the parity configuration -- by drawing, per image
    and per coordinate column, `k - d` values from torch's default CPU generator with `uniform_(min, max)` of the kept
    key points (bounds (`low`, min(image_size)) for an image without any).  This opt-in path (`pad_random: "torch_cpu"`)
    does exactly those draws in exactly that order, so that under the same `torch.manual_seed` the padded key points are
    the reference's, bit for bit.  It costs what the default (`pad_keypoints_native`: one launch, the library's own
    counter-based generator, no synchronisation) avoids: a host read of the counts and, for every image that needs
    padding, a copy of its key points to the host and back."""
    b, cap, _ = kpts.shape
    n = counts.tolist()  # host synchronisation
    high = float(data["image_size"].min().item()) if "image_size" in data else float(min(image.shape[-2:]))
    for i in range(b):
        d = min(int(n[i]), int(k))
        scores[i, d:] = 0.0  # pad_and_stack(scores, ..., mode="zeros")
        if d >= k:
            continue
        x = kpts[i, :d].cpu()
        cols = [torch.empty(k - d, 1).uniform_(float(x[:, c].min()) if d > 0 else float(low),
                                               float(x[:, c].max()) if d > 0 else high) for c in range(2)]
        kpts[i, d:k] = torch.cat(cols, -1).to(kpts.device)
    if cap == k:
        return kpts, scores
    return kpts[:, :k].contiguous(), scores[:, :k].contiguous()


class SuperPointRunner:
    """Launch sequence for one extractor call."""

    def __init__(self):
        self.ws = nat.Workspace()
        self.ws_sel = nat.Workspace()
        self.trace = None  # optional nat.KernelTrace (bench.py): per-launch events of the dominant kernel

    def dense(self, packed, image):
        lib = nat.lib()
        b, c, h, w = image.shape
        dev = image.device
        h8, w8 = h // 8, w // 8
        heat = torch.empty((b, h8 * 8, w8 * 8), device=dev, dtype=torch.float32)
        desc = torch.empty((b, h8, w8, packed.desc_dim), device=dev, dtype=torch.float32)
        need = lib.gfc_sp_workspace_bytes(b, c, h, w)
        ws = self.ws.get(need, dev)
        nat.check(lib.gfc_sp_dense(ctypes.byref(packed.params), nat.ptr(image), b, c, h, w, nat.ptr(heat),
                                   nat.ptr(desc), nat.ptr(ws), ws.numel(),
                                   ctypes.byref(self.trace.c) if self.trace is not None else None,
                                   nat.stream_ptr(dev)), "gfc_sp_dense")
        return heat, desc

    def nms(self, heat, radius, border, valid_wh=None):
        lib = nat.lib()
        b, h, w = heat.shape
        out = torch.empty_like(heat)
        nat.check(lib.gfc_sp_nms(nat.ptr(heat), b, h, w, int(radius), int(border), nat.ptr(valid_wh), nat.ptr(out),
                                 nat.stream_ptr(heat.device)), "gfc_sp_nms")
        return out

    def select(self, scores, threshold, k):
        """k: int >= 1, or None for unlimited.  Returns kpts [B,cap,2], kscores [B,cap], counts [B]."""
        lib = nat.lib()
        b, h, w = scores.shape
        dev = scores.device
        cap = int(k) if k is not None else h * w
        kpts = torch.empty((b, cap, 2), device=dev, dtype=torch.float32)
        ksc = torch.empty((b, cap), device=dev, dtype=torch.float32)
        counts = torch.empty((b,), device=dev, dtype=torch.int32)
        need = lib.gfc_sp_select_workspace_bytes(b, h, w)
        ws = self.ws_sel.get(need, dev)
        nat.check(lib.gfc_sp_select(nat.ptr(scores), b, h, w, float(threshold), int(k) if k is not None else -1, cap,
                                    nat.ptr(kpts), nat.ptr(ksc), nat.ptr(counts), nat.ptr(ws), ws.numel(),
                                    nat.stream_ptr(dev)), "gfc_sp_select")
        return kpts, ksc, counts

    def nms_select(self, heat, radius, border, valid_wh, threshold, k):
        """Fused NMS + top-k (finite k <= 8192): no dense suppressed map, no scan in the selection kernel."""
        lib = nat.lib()
        b, h, w = heat.shape
        dev = heat.device
        k = int(k)
        kpts = torch.empty((b, k, 2), device=dev, dtype=torch.float32)
        ksc = torch.empty((b, k), device=dev, dtype=torch.float32)
        counts = torch.empty((b,), device=dev, dtype=torch.int32)
        ws = self.ws_sel.get(lib.gfc_sp_nms_select_workspace_bytes(b, h, w), dev)
        nat.check(lib.gfc_sp_nms_select(nat.ptr(heat), b, h, w, int(radius), int(border), nat.ptr(valid_wh),
                                        float(threshold), k, k, None, nat.ptr(kpts), nat.ptr(ksc), nat.ptr(counts),
                                        nat.ptr(ws), ws.numel(), nat.stream_ptr(dev)), "gfc_sp_nms_select")
        return kpts, ksc, counts

    def refine_keypoints(self, heat, kpts, counts, radius):
        lib = nat.lib()
        b, h, w = heat.shape
        nat.check(lib.gfc_sp_refine_keypoints(nat.ptr(heat), b, h, w, nat.ptr(kpts), nat.ptr(counts), kpts.shape[1],
                                              int(radius), nat.stream_ptr(heat.device)), "gfc_sp_refine_keypoints")
        return kpts

    def mask_scores(self, scores, mask, image_wh):
        """Open-variant order: pixels outside the specular mask can no longer be selected (in place)."""
        lib = nat.lib()
        b, h, w = scores.shape
        nat.check(lib.gfc_sp_mask_scores(nat.ptr(scores), b, h, w, nat.ptr(mask), mask.shape[-2], mask.shape[-1],
                                         nat.ptr(image_wh), nat.stream_ptr(scores.device)), "gfc_sp_mask_scores")
        return scores

    def filter_keypoints(self, kpts, ksc, counts, mask, image_wh, offset=0.0):
        """Official-variant order: compact the selected key points that lie on the mask (in place, counts updated)."""
        lib = nat.lib()
        nat.check(lib.gfc_sp_filter_keypoints(nat.ptr(kpts), nat.ptr(ksc), nat.ptr(counts), kpts.shape[0], kpts.shape[1],
                                              nat.ptr(mask), mask.shape[-2], mask.shape[-1], nat.ptr(image_wh),
                                              float(offset), nat.stream_ptr(kpts.device)), "gfc_sp_filter_keypoints")
        return kpts, ksc, counts

    def sample(self, desc_raw, kpts, counts, mode):
        lib = nat.lib()
        b, h8, w8, d = desc_raw.shape
        cap = kpts.shape[1]
        out = torch.empty((b, cap, d), device=kpts.device, dtype=torch.float32)
        kout = torch.empty_like(kpts)
        nat.check(lib.gfc_sp_sample(nat.ptr(desc_raw), b, h8, w8, d, nat.ptr(kpts), nat.ptr(counts), cap, int(mode),
                                    nat.ptr(out), nat.ptr(kout), nat.stream_ptr(kpts.device)), "gfc_sp_sample")
        return out, kout

    def l2norm_rows(self, x):
        lib = nat.lib()
        rows = x.numel() // x.shape[-1]
        nat.check(lib.gfc_l2norm_rows(nat.ptr(x), rows, x.shape[-1], nat.stream_ptr(x.device)), "gfc_l2norm_rows")
        return x


def joint_pair_data(data0, data1):
    """Two single-view extractor inputs as ONE batch (view 0's images first), or None when they cannot share a call:
    different image shape / dtype / device, or side inputs (image_size, specular_mask) present in only one of them."""
    im0, im1 = data0["image"], data1["image"]
    if im0.shape[1:] != im1.shape[1:] or im0.dtype != im1.dtype or im0.device != im1.device:
        return None
    joint = {"image": torch.cat([im0, im1], 0)}
    for key in ("image_size", "specular_mask"):
        if (key in data0) != (key in data1):
            return None
        if key in data0:
            a, b = data0[key], data1[key]
            if a.shape[1:] != b.shape[1:]:
                return None
            joint[key] = torch.cat([a.to(im0.device), b.to(im0.device)], 0)
    return joint


class DeferredViews:
    """Per-image results of one extractor call whose key-point counts have not been read yet (run_extractor,
    defer_counts): `counts` [b] int32 on the device; `finish(lens)` -> the list of per-image prediction dicts."""

    def __init__(self, kout, ksc, desc, counts, core_ms, image, desc_raw, runner):
        self.kout, self.ksc, self.desc, self.counts = kout, ksc, desc, counts
        self.core_ms, self.image, self.desc_raw, self.runner = core_ms, image, desc_raw, runner

    def finish(self, lens):
        preds = []
        for i, n in enumerate(lens):
            p_i = {"keypoints": self.kout[i:i + 1, :n], "keypoint_scores": self.ksc[i:i + 1, :n],
                   "descriptors": self.desc[i:i + 1, :n],
                   "extractor_core_time_ms": self.image.new_full((1,), self.core_ms)}
            if self.desc_raw is not None:
                p_i["dense_descriptors"] = self.runner.l2norm_rows(self.desc_raw[i:i + 1].clone()).permute(0, 3, 1, 2)
            preds.append(p_i)
        return preds


def extract_views(model, views):
    """`[model(v) for v in views]` (each `v` a single-image extractor input) with ONE extractor call per distinct
    image shape among them: the views are grouped by what must agree inside a call (image shape / dtype / device, which
    side inputs are present and their shapes), each group runs as one batch with per-image key-point counts
    (`per_image=True`), and the predictions come back in the order given.  MI355X addition for the pair-batched
    evaluation loop (export_predictions(pair_batch=N)): at batch 1 the layers behind the stem fill a fraction of the
    chip.  Every image's result is what its own call returns (images of a batch are independent)."""
    groups = {}
    for i, v in enumerate(views):
        for key in model.required_data_keys:
            assert key in v, f"Missing key {key} in data"
        im = v["image"]
        sig = (tuple(im.shape[1:]), im.dtype, str(im.device), im.shape[0] == 1,
               tuple((k, tuple(v[k].shape[1:])) for k in ("image_size", "specular_mask") if k in v))
        groups.setdefault(sig, []).append(i)
    out = [None] * len(views)
    deferred = []
    shared = [idx for sig, idx in groups.items() if sig[3] and len(idx) > 1]
    # The weights are packed HERE, on the caller's stream, before any group is handed to a side stream: packing is a
    # sequence of kernels, and a lane that waited only on the caller's stream would otherwise read filters another lane
    # is still packing (round-4 finding: a never-run model whose first call was forward_views).
    for idx in shared:
        if views[idx[0]]["image"].device.type == "cuda":  # (anything else is refused by _forward: no CPU path)
            model.ensure_packed(views[idx[0]]["image"].device)
    # Two shape groups in flight: the groups are independent, so they alternate between two side streams (each with its
    # own runner = its own workspaces) and fill each other's ramps and tails (a group of ~13 VGA images is a small
    # batch for the chip).  Only with deferred counts (a finite max_num_keypoints, no padding): otherwise every group
    # ends in a host synchronisation and there is nothing to overlap.
    lanes = None
    if len(shared) > 1 and model.defers_counts():
        dev0 = views[shared[0][0]]["image"].device
        if dev0.type == "cuda" and all(views[idx[0]]["image"].device == dev0 for idx in shared):
            lanes = getattr(model, "_view_lanes", None)
            if lanes is None or lanes[0][0].device != dev0:
                lanes = [(torch.cuda.Stream(dev0), SuperPointRunner()) for _ in range(2)]
                model._view_lanes = lanes
    main = torch.cuda.current_stream(dev0) if lanes else None
    used = set()
    n_shared = 0
    for sig, idx in groups.items():
        if not sig[3] or len(idx) == 1:  # batched views (or a shape of its own): the ordinary call
            for i in idx:
                out[i] = model(views[i])
            continue
        dev = views[idx[0]]["image"].device

        def run_group(runner):
            joint = {"image": torch.cat([views[i]["image"] for i in idx], 0)}
            for key in ("image_size", "specular_mask"):
                if key in views[idx[0]]:
                    joint[key] = torch.cat([views[i][key].to(dev) for i in idx], 0)
            return model._forward(joint, per_image=True, defer_counts=True, runner=runner)

        if lanes:
            stream, runner = lanes[n_shared % 2]
            n_shared += 1
            if id(stream) not in used:
                stream.wait_stream(main)  # the views' tensors were produced on the caller's stream
                used.add(id(stream))
            with torch.cuda.stream(stream):
                res = run_group(runner)
            if not isinstance(res, DeferredViews):  # (unlimited key points: per-image results with a host sync; rare)
                main.wait_stream(stream)
        else:
            res = run_group(None)
        if isinstance(res, DeferredViews):
            deferred.append((idx, res))
        else:
            for i, pred in zip(idx, res):
                out[i] = pred
    if lanes:
        for stream, _ in lanes:
            if id(stream) in used:
                main.wait_stream(stream)
        for _, res in deferred:  # allocated on a side stream, consumed on the caller's from here on
            for t in (res.kout, res.ksc, res.desc, res.counts, res.desc_raw):
                if t is not None:
                    t.record_stream(main)
    if deferred:
        # ONE host synchronisation for the key-point counts of all shape groups (the groups' kernels queue up behind
        # each other meanwhile) instead of one per extractor call
        lens = torch.cat([res.counts for _, res in deferred]).tolist()
        o = 0
        for idx, res in deferred:
            for i, pred in zip(idx, res.finish(lens[o:o + len(idx)])):
                out[i] = pred
            o += len(idx)
    return out


def specular_mask_bytes(data, b, device):
    """`data["specular_mask"]` ([B,1,H,W] / [B,H,W], any dtype, non-zero = keep; extractors/utils.py:16-20) as
    contiguous bytes [B,Hm,Wm] on the device, and image_size as int32 [B,2] (w, h) or None."""
    m = data["specular_mask"]
    m = m.reshape((b,) + tuple(m.shape[-2:])).to(device)
    m = (m if m.dtype == torch.bool else m != 0).to(torch.uint8).contiguous()
    wh = None
    if data.get("image_size") is not None:
        # `int(w)` truncates towards zero (utils.py:15-16)
        wh = data["image_size"].to(device=device).reshape(b, 2).to(torch.int32).contiguous()
    return m, wh


def run_extractor(runner, packed, data, *, nms_radius, remove_borders, detection_threshold, max_num_keypoints,
                  force_num_keypoints, sample_mode, use_image_size_for_borders, dense_outputs, specular=None,
                  refinement_radius=0, per_image=False, defer_counts=False, pad_random="device"):
    """Shared `_forward` body (superpoint_open.py:126-232 / superpoint.py:206-379).
    specular: None, "before_topk" (superpoint_open.py:177-188) or "after_topk" (superpoint.py:310-328) when
    `data["specular_mask"]` is to be applied.
    per_image: return a list of one prediction dict per image (batch dimension 1 each) instead of one batched dict;
    images may then yield different numbers of key points (two views of a pair extracted by ONE call,
    two_view_pipeline.py: the reference runs the extractor once per view, so its views never had to agree).
    defer_counts (with per_image, a finite max_num_keypoints and no padding): no host synchronisation here -- every
    image's slots are sampled up to the cap (beyond its count: zeros) and a `DeferredViews` is returned, whose
    `finish(lens)` cuts the per-image predictions once the caller has read the counts (extract_views reads the counts of
    ALL its shape groups with one synchronisation instead of one per extractor call)."""
    image = data["image"]
    nat.require_cuda(image, "data['image']")
    if image.dtype != torch.float32:
        image = image.float()
    image = image.contiguous()
    b = image.shape[0]
    core_start = time.perf_counter()
    heat, desc_raw = runner.dense(packed, image)
    valid_wh = None
    if use_image_size_for_borders and "image_size" in data and remove_borders:
        valid_wh = data["image_size"].to(device=image.device).to(torch.int32).contiguous()
    k = max_num_keypoints
    smask = smask_wh = None
    if specular is not None:
        smask, smask_wh = specular_mask_bytes(data, b, image.device)
    if specular != "before_topk" and k is not None and 0 < k <= 8192 and nms_radius >= 1:
        kpts, ksc, counts = runner.nms_select(heat, nms_radius, remove_borders or 0, valid_wh, detection_threshold, k)
        core_time_ms = (time.perf_counter() - core_start) * 1e3  # like the reference: no device sync
    else:  # unlimited number of key points: dense suppressed map + ordered scan
        suppressed = runner.nms(heat, nms_radius, remove_borders or 0, valid_wh)
        core_time_ms = (time.perf_counter() - core_start) * 1e3
        if specular == "before_topk":
            runner.mask_scores(suppressed, smask, smask_wh)
        kpts, ksc, counts = runner.select(suppressed, detection_threshold, k)
    if refinement_radius and refinement_radius > 0:  # superpoint.py:302-305: after top-k, before the specular filter
        runner.refine_keypoints(heat, kpts, counts, refinement_radius)
    if specular == "after_topk":
        runner.filter_keypoints(kpts, ksc, counts, smask, smask_wh, 0.0)
    if force_num_keypoints:
        if k is None:
            raise ValueError("force_num_keypoints needs max_num_keypoints")
        if pad_random == "torch_cpu":  # the reference's own random numbers (host round trip: parity runs)
            kpts, ksc = pad_keypoints_torch_cpu(kpts, ksc, counts, k, 0, data, image)
        elif pad_random == "device":  # kept on the device: no host synchronisation on the batched path, one launch
            kpts, ksc = pad_keypoints_native(kpts, ksc, counts, k, 0, data, image)
        else:
            raise ValueError(f"pad_random {pad_random!r}: 'device' or 'torch_cpu'")
        counts_arg = None
    elif defer_counts and per_image and k is not None:
        desc, kout = runner.sample(desc_raw, kpts, counts, sample_mode)
        return DeferredViews(kout, ksc, desc, counts, core_time_ms / b, image,
                             desc_raw if dense_outputs else None, runner)
    else:
        n = counts.tolist()  # host sync, as torch.where in the reference
        if len(set(n)) != 1 and not per_image:
            # the reference cannot stack ragged key-point lists either (torch.stack raises)
            raise RaggedCounts(f"images of one batch yield different numbers of keypoints {n}: "
                               "use force_num_keypoints=True or batch size 1")
        if len(set(n)) == 1:
            if n[0] != kpts.shape[1]:
                kpts, ksc = kpts[:, : n[0]].contiguous(), ksc[:, : n[0]].contiguous()
            counts_arg = None
        else:
            # ragged (per_image): the sampler zero-fills the slots beyond each image's count.  Trimmed to the largest
            # count first: with max_num_keypoints None the selection has H*W slots per image, and a [b, H*W, 256]
            # descriptor tensor (630 MB per VGA pair) would be allocated, zero-filled and kept alive by the views
            nmax = max(n)
            if nmax != kpts.shape[1]:
                kpts, ksc = kpts[:, :nmax].contiguous(), ksc[:, :nmax].contiguous()
            counts_arg = counts
    if kpts.shape[1] > 0:
        desc, kout = runner.sample(desc_raw, kpts, counts_arg, sample_mode)
    else:
        desc = kpts.new_zeros((b, 0, packed.desc_dim))
        kout = kpts
    if per_image:
        lens = n if (not force_num_keypoints and len(set(n)) != 1) else [kout.shape[1]] * b
        preds = []
        for i in range(b):
            # full-length rows stay views of the batched tensors (adjacent in memory: the matcher reads both views'
            # rows without a copy); shorter rows are trimmed
            p_i = {"keypoints": kout[i:i + 1, : lens[i]], "keypoint_scores": ksc[i:i + 1, : lens[i]],
                   "descriptors": desc[i:i + 1, : lens[i]],
                   "extractor_core_time_ms": image.new_full((1,), core_time_ms / b)}
            if dense_outputs:
                p_i["dense_descriptors"] = runner.l2norm_rows(desc_raw[i:i + 1].clone()).permute(0, 3, 1, 2)
            preds.append(p_i)
        return preds
    pred = {
        "keypoints": kout,
        "keypoint_scores": ksc,
        "descriptors": desc,
        "extractor_core_time_ms": image.new_full((b,), core_time_ms / b),
    }
    if dense_outputs:
        dense = runner.l2norm_rows(desc_raw.clone())
        pred["dense_descriptors"] = dense.permute(0, 3, 1, 2)  # [B,C,h,w] view of the NHWC map
    return pred

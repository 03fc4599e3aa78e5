"""Two-view pipeline: extractor on both views, key suffixing, matcher, timing keys.

Counterpart of gluefactory/models/two_view_pipeline.py:278-405 (`TwoViewPipeline._forward`)
for the components that are in scope (extractor + matcher; filter / solver / ground truth /
key-point rotation augmentation are outside the hot path and rejected).  The real
`gluefactory.models.two_view_pipeline.TwoViewPipeline` works unchanged with this package's
modules selected by name; this look-alike exists because the reference does not travel to the
GPU box.  Convention (two_view_pipeline.py:9-10): m0[i] = index in image 1 matched to keypoint
i of image 0, -1 if unmatched.
"""
import time

import torch

from .base_model import BaseModel, conf_get, to_plain
from .registry import get_model


class TwoViewPipeline(BaseModel):
    default_conf = {
        "extractor": {"name": None, "trainable": False},
        "matcher": {"name": None},
        "filter": {"name": None},
        "solver": {"name": None},
        "ground_truth": {"name": None},
        "allow_no_extract": False,
        "run_gt_in_forward": False,
        # random key-point rotation of one view (two_view_pipeline.py:38-43,161-276): a training augmentation
        # ("train_only"), never active on the inference path; asking for it at inference time is rejected
        "keypoint_rotation": {"enabled": False, "max_deg": 180.0, "view": 0, "train_only": True},
        # MI355X addition: extract both views with ONE extractor call when their images agree in shape and the
        # extractor offers `forward_pair` (same results; `extractor_time_ms` is then the time of that one call)
        "joint_extraction": True,
        # MI355X addition: False skips the reference's per-call profiling (device synchronisation before and after the
        # extractor and the matcher plus the peak-memory bookkeeping, two_view_pipeline.py:78-102) and with it the
        # optional timing / memory keys of the prediction: the pair then runs without a host stall between its stages
        "profile_calls": True,
    }
    required_data_keys = ["view0", "view1"]
    strict_conf = False
    components = ["extractor", "matcher", "filter", "solver", "ground_truth"]

    def _init(self, conf):
        for comp in ("filter", "solver", "ground_truth"):
            if conf_get(conf_get(conf, comp, {}), "name"):
                raise NotImplementedError(f"pipeline component {comp!r} is outside the accelerated hot path")
        rot = conf_get(conf, "keypoint_rotation", {})
        if conf_get(rot, "enabled") and not conf_get(rot, "train_only", True):
            raise NotImplementedError("keypoint_rotation outside training (a random augmentation) is not built")
        ext = conf_get(conf, "extractor")
        if conf_get(ext, "name"):
            self.extractor = get_model(conf_get(ext, "name"))(to_plain(ext))
        mat = conf_get(conf, "matcher")
        if conf_get(mat, "name"):
            self.matcher = get_model(conf_get(mat, "name"))(to_plain(mat))

    def is_initialized(self):
        ok = True
        for name in ("extractor", "matcher"):
            m = getattr(self, name, None)
            if m is not None:
                ok = ok and bool(m.is_initialized())
        return ok

    def _timed(self, device, fn):
        """two_view_pipeline.py:78-102: device-synchronised wall clock + peak-memory delta."""
        mem = None
        if not conf_get(self.conf, "profile_calls", True):
            return fn(), None, None
        if device.type == "cuda":
            # the calling thread's stream (= the whole device in the reference's single-stream use); several export
            # workers time their own pairs without serialising each other (their memory figures then overlap)
            torch.cuda.current_stream(device).synchronize()
            baseline = torch.cuda.memory_allocated(device)
            torch.cuda.reset_peak_memory_stats(device)
        start = time.perf_counter()
        out = fn()
        if device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()
            mem = max(torch.cuda.max_memory_allocated(device) - baseline, 0) / (1024 ** 2)
        return out, (time.perf_counter() - start) * 1e3, mem

    def extract_view(self, data, i):
        data_i = data[f"view{i}"]
        pred_i = dict(data_i.get("cache", {}))
        skip = len(pred_i) > 0 and conf_get(self.conf, "allow_no_extract")
        t_ms = core_ms = mem = None
        if hasattr(self, "extractor") and not skip:
            inp = data_i if not pred_i else {**data_i, **pred_i}
            out, t_ms, mem = self._timed(data_i["image"].device, lambda: self.extractor(inp))
            core_ms = out.pop("extractor_core_time_ms", None)
            pred_i = {**pred_i, **out}
        return pred_i, t_ms, core_ms, mem

    def extract_pair(self, data):
        """Both views through one extractor call (extractors with `forward_pair`; no cached features involved).
        None when that does not apply: the caller then extracts the views one after the other like the reference
        (two_view_pipeline.py:283-284)."""
        ext = getattr(self, "extractor", None)
        d0, d1 = data["view0"], data["view1"]
        if ext is None or not hasattr(ext, "forward_pair") or d0.get("cache") or d1.get("cache"):
            return None
        if d0["image"].shape[1:] != d1["image"].shape[1:]:
            return None
        (out0, out1), t_ms, mem = self._timed(d0["image"].device, lambda: ext.forward_pair(d0, d1))
        c0, c1 = out0.pop("extractor_core_time_ms", None), out1.pop("extractor_core_time_ms", None)
        return out0, out1, t_ms, c0, c1, mem

    def _forward(self, data):
        image0, image1 = data["view0"]["image"], data["view1"]["image"]
        device = image0.device
        joint = self.extract_pair(data) if conf_get(self.conf, "joint_extraction", True) else None
        if joint is not None:
            pred0, pred1, t0, c0, c1, mem0 = joint
            t1 = mem1 = None
        else:
            pred0, t0, c0, mem0 = self.extract_view(data, "0")
            pred1, t1, c1, mem1 = self.extract_view(data, "1")
        pred = {**{k + "0": v for k, v in pred0.items()}, **{k + "1": v for k, v in pred1.items()}}
        t_match = mem_match = None
        if hasattr(self, "matcher"):
            out, t_match, mem_match = self._timed(device, lambda: self.matcher({**data, **pred}))
            pred = {**pred, **out}
        return self._with_timing_keys(pred, image0, image1, (t0, t1), (c0, c1), (mem0, mem1), t_match, mem_match)

    def _with_timing_keys(self, pred, image0, image1, ext_t, ext_core, ext_mem, t_match, mem_match):
        """The timing / memory / resolution keys of two_view_pipeline.py:286-339 on a pair's prediction."""
        device, b = image0.device, image0.shape[0]
        (t0, t1), (c0, c1), (mem0, mem1) = ext_t, ext_core, ext_mem

        def full(v):
            return torch.full((b,), float(v), device=device, dtype=torch.float32)

        ext_times = [t for t in (t0, t1) if t is not None]
        if ext_times:
            pred["extractor_time_ms"] = full(sum(ext_times))
            cores = [c for c in (c0, c1) if c is not None]
            if cores:
                pred["extractor_core_time_ms"] = sum(cores)
            total = sum(ext_times)
            if t_match is not None:
                pred["matcher_time_ms"] = full(t_match)
                total += t_match
            pred["total_time_ms"] = full(total)
        elif t_match is not None:
            pred["total_time_ms"] = full(t_match)
        ext_mem = [m for m in (mem0, mem1) if m is not None]
        if ext_mem:
            pred["extractor_memory_mb"] = full(sum(ext_mem))
        if mem_match is not None:
            pred["matcher_memory_mb"] = full(mem_match)
        pred["pair_resolution"] = full(image0.shape[-2] * image0.shape[-1] + image1.shape[-2] * image1.shape[-1])
        return pred

    def forward_pairs(self, datas, view_keys=None):
        """MI355X addition: `[self(d) for d in datas]` for batch-1 pairs whose IMAGES DIFFER IN SIZE (the HPatches
        evaluation list: datasets/hpatches.py:60 asserts batch size 1, utils/export_predictions.py:36-45 runs the model
        pair by pair) with both stages batched: the extractor once per distinct image shape among the 2N views
        (`forward_views`), the matcher ONCE over all N pairs with their own key-point counts (`forward_pairs`,
        gfc_lg_forward_ragged).  Every pair's prediction carries the keys of the single-pair call; the timing /
        memory keys are the batch's figures divided by N (one device-synchronised measurement per stage and batch).
        Pairs with cached features, batched pairs, or a matcher without `forward_pairs` take `self(d)`.
        view_keys (optional): one (key0, key1) per pair; views with EQUAL non-None keys are declared by the caller to be
        the same image and are extracted once (an HPatches sequence reads its reference image as view 0 of all five of its
        pairs, datasets/hpatches.py:98-99: 32 consecutive pairs hold ~39 distinct images, not 64).  Every pair's
        prediction is what it is without the keys -- an image's features do not depend on what else is in its batch."""
        ext, mat = getattr(self, "extractor", None), getattr(self, "matcher", None)
        ok = len(datas) > 1 and ext is not None and (mat is None or hasattr(mat, "forward_pairs"))
        for d in datas:
            for key in self.required_data_keys:
                assert key in d, f"Missing key {key} in data"
            ok = ok and not d["view0"].get("cache") and not d["view1"].get("cache") \
                and d["view0"]["image"].shape[0] == 1 and d["view1"]["image"].shape[0] == 1
        if not ok:
            return [self(d) for d in datas]
        n = len(datas)
        device = datas[0]["view0"]["image"].device
        views = [d[f"view{i}"] for d in datas for i in ("0", "1")]
        # extractors without `forward_views` (DISK: its own chunked batching) run view by view; the matcher is batched either way
        many = ext.forward_views if hasattr(ext, "forward_views") else (lambda vs: [ext(v) for v in vs])
        slot = list(range(len(views)))  # view -> index of the extraction that serves it
        if view_keys is not None:
            assert len(view_keys) == n, "one (key0, key1) per pair"
            flat = [k for pair in view_keys for k in pair]
            first, uniq = {}, []
            for j, (v, k) in enumerate(zip(views, flat)):
                if k is not None and k in first:
                    u = first[k]
                    if uniq[u]["image"].shape != v["image"].shape:
                        raise ValueError(f"view key {k!r} names images of different shapes "
                                         f"{tuple(uniq[u]['image'].shape)} / {tuple(v['image'].shape)}")
                else:
                    u = len(uniq)
                    uniq.append(v)
                    if k is not None:
                        first[k] = u
                slot[j] = u
            views = uniq
        vpreds, t_ext, mem_ext = self._timed(device, lambda: many(views))
        vpreds = [dict(vpreds[u]) for u in slot]
        cores = [vp.pop("extractor_core_time_ms", None) for vp in vpreds]
        preds = []
        for j, d in enumerate(datas):
            p0, p1 = vpreds[2 * j], vpreds[2 * j + 1]
            preds.append({**{k + "0": v for k, v in p0.items()}, **{k + "1": v for k, v in p1.items()}})
        t_match = mem_match = None
        if mat is not None:
            outs, t_match, mem_match = self._timed(device, lambda: mat.forward_pairs(
                [{**d, **p} for d, p in zip(datas, preds)]))
            preds = [{**p, **o} for p, o in zip(preds, outs)]

        def share(v):
            return None if v is None else v / n

        return [self._with_timing_keys(p, d["view0"]["image"], d["view1"]["image"], (share(t_ext), None),
                                       (cores[2 * j], cores[2 * j + 1]), (share(mem_ext), None), share(t_match),
                                       share(mem_match))
                for j, (d, p) in enumerate(zip(datas, preds))]

    def loss(self, pred, data):
        raise NotImplementedError("training is out of scope")


__main_model__ = TwoViewPipeline

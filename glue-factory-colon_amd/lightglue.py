"""LightGlue matcher on MI355X -- drop-in for `gluefactory.models.matchers.lightglue`
(reference file gluefactory/models/matchers/lightglue.py:322-640, inference path).

Same configuration keys, same `forward(data) -> dict` contract (lightglue.py:422-553) and
the same state-dict key names (lightglue.py:349-408), including the legacy
`self_attn.{i}` -> `transformers.{i}.self_attn` rename (lightglue.py:394-401), so the official
`superpoint_lightglue.pth` / `disk_lightglue.pth` load unchanged.  Like the reference class it
is a plain nn.Module (not a BaseModel) resolved through `__main_model__`.  The torch
sub-modules are parameter containers only; the forward pass is one call into libgfc_amd.so.

    model.matcher.name = glue_factory_colon_amd.lightglue
"""
import ctypes
import sys
import threading
from pathlib import Path

import torch
from torch import nn

from . import _native as nat
from . import weights as _weights
from .base_model import Conf, conf_get, merge


def _ffn(d):
    return nn.Sequential(nn.Linear(2 * d, 2 * d), nn.LayerNorm(2 * d, elementwise_affine=True), nn.GELU(),
                         nn.Linear(2 * d, d))


class _SelfBlock(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.Wqkv = nn.Linear(d, 3 * d)
        self.out_proj = nn.Linear(d, d)
        self.ffn = _ffn(d)


class _CrossBlock(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.to_qk = nn.Linear(d, d)
        self.to_v = nn.Linear(d, d)
        self.to_out = nn.Linear(d, d)
        self.ffn = _ffn(d)


class _Layer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.self_attn = _SelfBlock(d)
        self.cross_attn = _CrossBlock(d)


class _PosEnc(nn.Module):
    def __init__(self, m, f_dim):
        super().__init__()
        self.Wr = nn.Linear(m, f_dim // 2, bias=False)


class _Assignment(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.matchability = nn.Linear(d, 1)
        self.final_proj = nn.Linear(d, d)


class _TokenConfidence(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.token = nn.Sequential(nn.Linear(d, 1), nn.Sigmoid())


_CAPTURE_LOCK = threading.Lock()  # one stream capture at a time per process


def _pack_sides(a, b):
    """[B,M,C] and [B,N,C] -> [B*M + B*N, C] rows (side 0 first).  A view when `b` starts where `a` ends in the same
    allocation (both views extracted by one call), a concatenation otherwise."""
    c = a.shape[-1]
    ra, rb = a.numel() // c, b.numel() // c
    if (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype == torch.float32
            and b.data_ptr() == a.data_ptr() + a.numel() * 4
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()):
        return a.as_strided((ra + rb, c), (c, 1))
    return torch.cat([a.reshape(ra, c), b.reshape(rb, c)], 0)


class LightGlue(nn.Module):
    default_conf = {
        "name": "lightglue",
        "input_dim": 256,
        "add_scale_ori": False,
        "descriptor_dim": 256,
        "n_layers": 9,
        "num_heads": 4,
        "flash": False,
        "mp": False,
        "depth_confidence": -1,
        "width_confidence": -1,
        "filter_threshold": 0.0,
        "checkpointed": False,
        "weights": None,  # path of a checkpoint, or "synthetic[:seed]" for the name-seeded weights
        "weights_from_version": "v0.1_arxiv",
        "loss": {"gamma": 1.0, "fn": "nll", "nll_balancing": 0.5},
        # MI355X-specific: fold out_proj / to_out into the first FFN matrix at load time (one GEMM and one
        # [rows,256] HBM round trip less per block; same function, rounding differs by ~1e-7 relative)
        "fold_out_proj": True,
        # MI355X-specific, opt-in: problems of at most this many rows (b * (m + n); batch-1 evaluation: 2048) replay the
        # matcher's ~100 launches as ONE captured HIP graph per (b, m, n) (static buffers, inputs copied in, outputs
        # copied out).  Same kernels, same results.  0 (default) = eager launches: measured at batch 1 the graph saves
        # 0.05 of 1.65 ms per pair (the matcher is bound by its small grids, not by launch gaps),
        # and HIP refuses concurrent captures from several host threads (export_predictions(workers > 1)).
        "graph_max_rows": 0,
    }
    required_data_keys = ["keypoints0", "keypoints1", "descriptors0", "descriptors1"]

    def __init__(self, conf) -> None:
        super().__init__()
        self.conf = conf = Conf(merge(self.default_conf, conf))
        if conf.descriptor_dim != 256 or conf.num_heads != 4:
            raise NotImplementedError("the MI355X kernels are built for descriptor_dim 256, 4 heads (head_dim 64)")
        if conf.n_layers > nat.GFC_LG_MAX_LAYERS:
            raise NotImplementedError(f"at most {nat.GFC_LG_MAX_LAYERS} layers")
        if conf.input_dim != conf.descriptor_dim:
            if conf.input_dim % 32:
                raise NotImplementedError("input_dim must be a multiple of 32")
            self.input_proj = nn.Linear(conf.input_dim, conf.descriptor_dim, bias=True)
        else:
            self.input_proj = nn.Identity()
        d, n = conf.descriptor_dim, conf.n_layers
        head_dim = d // conf.num_heads
        self.posenc = _PosEnc(2 + 2 * bool(conf.add_scale_ori), head_dim)  # lightglue.py:358-360
        self.transformers = nn.ModuleList([_Layer(d) for _ in range(n)])
        self.log_assignment = nn.ModuleList([_Assignment(d) for _ in range(n)])
        self.token_confidence = nn.ModuleList([_TokenConfidence(d) for _ in range(n - 1)])
        self.register_buffer("confidence_thresholds", _weights.confidence_thresholds(n))
        self._packed = None
        self._ws = nat.Workspace()
        self.trace = None  # optional nat.KernelTrace (bench.py): per-launch events of the attention kernel
        self._graphs = {}  # (b, m, n, device, has scale/ori) -> captured launch sequence + its static buffers
        self.are_weights_initialized = False

        w = conf.weights
        if w is not None:
            if Path(str(w)).exists():
                self.load_state_dict(torch.load(str(w), map_location="cpu"), strict=False)
            elif isinstance(w, str) and w.startswith("synthetic"):
                seed = int(w.split(":")[1]) if ":" in w else 0
                self.load_state_dict(_weights.lightglue_state_dict(seed, input_dim=conf.input_dim, n_layers=n),
                                     strict=False)
            else:
                # the reference downloads `{weights}_lightglue.pth` here (lightglue.py:385-392); no network
                raise FileNotFoundError(f"weights {w!r} not found (no download is attempted)")

    # -- weights ------------------------------------------------------------------------
    def load_state_dict(self, state_dict, *args, **kwargs):
        for i in range(self.conf.n_layers):  # legacy key names, lightglue.py:394-401
            state_dict = {k.replace(f"self_attn.{i}", f"transformers.{i}.self_attn"): v for k, v in state_dict.items()}
            state_dict = {k.replace(f"cross_attn.{i}", f"transformers.{i}.cross_attn"): v
                          for k, v in state_dict.items()}
        ret = super().load_state_dict(state_dict, *args, **kwargs)
        self._packed = None
        self.are_weights_initialized = True
        return ret

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        self._graphs = {}
        return super()._apply(fn, *args, **kwargs)

    def is_initialized(self):
        return self.are_weights_initialized

    def __deepcopy__(self, memo):
        """Replicas (export_predictions workers) get their own parameters and workspaces; captured graphs and packed
        weight pointers belong to the original and are rebuilt by the copy on first use."""
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_graphs":
                new.__dict__[k] = {}
            elif k == "_packed":
                new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    # -- small problems: the whole launch sequence as one HIP graph ------------------------
    def _launch_packed(self, kp, de, s0, s1, so, b, m, n, m0, m1, ms0, ms1, scores, rows, ws, trace=None):
        lib = nat.lib()
        nat.check(lib.gfc_lg_forward_packed(
            ctypes.byref(self._packed[0]), nat.ptr(kp), nat.ptr(de), nat.ptr(s0), nat.ptr(s1), nat.ptr(so), b, m, n,
            float(self.conf.filter_threshold), nat.ptr(m0), nat.ptr(m1), nat.ptr(ms0), nat.ptr(ms1), nat.ptr(scores),
            nat.ptr(rows), nat.ptr(ws), ws.numel(), ctypes.byref(trace.c) if trace is not None else None,
            nat.stream_ptr(kp.device)), "gfc_lg_forward_packed")

    def _graph_entry(self, key, kp, de, s0, s1, so, b, m, n):
        """Static buffers + the captured launch sequence of gfc_lg_forward_packed for one problem shape."""
        device, d = kp.device, self.conf.descriptor_dim
        lib = nat.lib()
        e = {"kp": torch.empty_like(kp), "de": torch.empty_like(de), "s0": torch.empty_like(s0),
             "s1": torch.empty_like(s1), "so": None if so is None else torch.empty_like(so),
             "m0": torch.empty((b, m), device=device, dtype=torch.long),
             "m1": torch.empty((b, n), device=device, dtype=torch.long),
             "ms0": torch.empty((b, m), device=device), "ms1": torch.empty((b, n), device=device),
             "scores": torch.empty((b, m + 1, n + 1), device=device),
             "rows": torch.empty((b * (m + n), d), device=device),
             "ws": torch.empty(int(lib.gfc_lg_packed_workspace_bytes(b, m, n)), dtype=torch.uint8, device=device)}
        for name, src in (("kp", kp), ("de", de), ("s0", s0), ("s1", s1), ("so", so)):
            if src is not None:
                e[name].copy_(src)

        def run():
            self._launch_packed(e["kp"], e["de"], e["s0"], e["s1"], e["so"], b, m, n, e["m0"], e["m1"], e["ms0"],
                                e["ms1"], e["scores"], e["rows"], e["ws"])

        # one eager run first (per-kernel attributes are set on first use), then the capture
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream(device).wait_stream(side)
        try:
            with _CAPTURE_LOCK:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    run()
            e["graph"] = graph
        except (nat.NativeError, RuntimeError) as exc:
            # RuntimeError: the capture was refused (another thread is using the device).  NativeError: raised by run()
            # inside the `with` block, i.e. always while this stream was capturing -- the eager run above has just
            # proved the same launch sequence on the same buffers, so the failure is specific to capture.  Either way
            # this shape runs with eager launches from now on (the caller does that run: nothing is repeated here), and
            # the reason is logged once per kind, never hidden.
            kind = "launch failed under capture" if isinstance(exc, nat.NativeError) else "capture refused"
            logged = LightGlue.__dict__.get("_graph_fallback_logged") or set()
            if kind not in logged:
                LightGlue._graph_fallback_logged = logged | {kind}
                print(f"glue_factory_colon_amd.lightglue: HIP graph {kind} ({exc}); problems of shape "
                      f"{key[:3]} run with eager launches", file=sys.stderr)
            e = {"graph": None}
        self._graphs[key] = e
        return e

    def _pack(self, device):
        conf = self.conf
        keep = []

        def dev(t):
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        p = nat.LgParams()
        p.n_layers, p.input_dim = conf.n_layers, conf.input_dim
        if conf.input_dim != conf.descriptor_dim:
            p.input_proj_w, p.input_proj_b = dev(self.input_proj.weight), dev(self.input_proj.bias)
        p.posenc_wr = dev(self.posenc.Wr.weight)
        p.posenc_dim = 4 if self.conf.add_scale_ori else 2
        d, h = conf.descriptor_dim, conf.num_heads
        dh = d // h
        # Wqkv rows: state-dict row = head*(3*dh) + dd*3 + s  ->  packed row = s*d + head*dh + dd
        idx = torch.arange(3 * d)
        s_, rem = idx // d, idx % d
        head, dd = rem // dh, rem % dh
        src = head * (3 * dh) + dd * 3 + s_
        fold = bool(conf.fold_out_proj)

        def ffn0(lin0, out):
            """ffn[0] weights with `out` (out_proj / to_out) folded into the message half when enabled."""
            if not fold:
                return lin0.weight, lin0.bias
            w0, b0 = lin0.weight.detach().double(), lin0.bias.detach().double()
            wo, bo = out.weight.detach().double(), out.bias.detach().double()
            w = torch.cat([w0[:, :d], w0[:, d:] @ wo], 1)
            return w.float(), (b0 + w0[:, d:] @ bo).float()

        for i, layer in enumerate(self.transformers):
            sa, ca = layer.self_attn, layer.cross_attn
            p.wqkv[i] = dev(sa.Wqkv.weight[src])
            p.bqkv[i] = dev(sa.Wqkv.bias[src])
            if not fold:
                p.s_out_w[i], p.s_out_b[i] = dev(sa.out_proj.weight), dev(sa.out_proj.bias)
                p.c_out_w[i], p.c_out_b[i] = dev(ca.to_out.weight), dev(ca.to_out.bias)
            w0, b0 = ffn0(sa.ffn[0], sa.out_proj)
            p.s_ffn0_w[i], p.s_ffn0_b[i] = dev(w0), dev(b0)
            p.s_ln_g[i], p.s_ln_b[i] = dev(sa.ffn[1].weight), dev(sa.ffn[1].bias)
            p.s_ffn3_w[i], p.s_ffn3_b[i] = dev(sa.ffn[3].weight), dev(sa.ffn[3].bias)
            p.c_qkv_w[i] = dev(torch.cat([ca.to_qk.weight, ca.to_v.weight], 0))
            p.c_qkv_b[i] = dev(torch.cat([ca.to_qk.bias, ca.to_v.bias], 0))
            w0, b0 = ffn0(ca.ffn[0], ca.to_out)
            p.c_ffn0_w[i], p.c_ffn0_b[i] = dev(w0), dev(b0)
            p.c_ln_g[i], p.c_ln_b[i] = dev(ca.ffn[1].weight), dev(ca.ffn[1].bias)
            p.c_ffn3_w[i], p.c_ffn3_b[i] = dev(ca.ffn[3].weight), dev(ca.ffn[3].bias)
        for i, head in enumerate(self.log_assignment):
            p.final_proj_w[i], p.final_proj_b[i] = dev(head.final_proj.weight), dev(head.final_proj.bias)
            p.matchability_w[i] = dev(head.matchability.weight.reshape(-1))
            p.matchability_b[i] = dev(head.matchability.bias)
        for i, tc in enumerate(self.token_confidence):
            p.token_w[i], p.token_b[i] = dev(tc.token[0].weight.reshape(-1)), dev(tc.token[0].bias)
        return p, keep, device

    def ensure_packed(self, device):
        """The device copies of the weights in the library's layouts, built on the CALLING thread's current stream if
        they do not exist yet (export workers share them: the caller packs before its worker streams start)."""
        if not self.are_weights_initialized:
            raise RuntimeError("LightGlue weights are not loaded (conf.weights or load_state_dict)")
        if self._packed is None or self._packed[2] != device:
            self._packed = self._pack(device)
            self._graphs = {}
        return self._packed

    # -- forward ------------------------------------------------------------------------
    def forward(self, data: dict) -> dict:
        for key in self.required_data_keys:
            assert key in data, f"Missing key {key} in data"
        conf = self.conf
        if self.training:
            raise NotImplementedError("training (loss, checkpointing) is out of scope: inference path only")
        if not self.are_weights_initialized:
            raise RuntimeError("LightGlue weights are not loaded (conf.weights or load_state_dict)")
        kpts0, kpts1 = data["keypoints0"], data["keypoints1"]
        nat.require_cuda(kpts0, "data['keypoints0']")
        b, m, _ = kpts0.shape
        b, n, _ = kpts1.shape
        device = kpts0.device
        # like the reference (lightglue.py:430-434) the image sizes come from the views; a view without
        # `image_size` normalises by the extent of its own key points (normalize_keypoints, lightglue.py:31-32).
        # (The reference leaves size0/size1 unbound when a view is absent; here that is treated like a missing size.)
        size0 = data.get("view0", {}).get("image_size")
        size1 = data.get("view1", {}).get("image_size")
        if size0 is None and m > 0:
            size0 = 1 + kpts0.float().amax(-2) - kpts0.float().amin(-2)
        if size1 is None and n > 0:
            size1 = 1 + kpts1.float().amax(-2) - kpts1.float().amin(-2)
        desc0 = data["descriptors0"].contiguous().float()
        desc1 = data["descriptors1"].contiguous().float()
        assert desc0.shape[-1] == conf.input_dim
        assert desc1.shape[-1] == conf.input_dim
        so0 = so1 = None
        if conf.add_scale_ori:  # lightglue.py:436-453: [x, y, scale, orientation] feeds the positional encoding
            def pack(sc, ori):
                sc = sc if sc.dim() == 3 else sc[..., None]
                ori = ori if ori.dim() == 3 else ori[..., None]
                return torch.cat([sc, ori], -1).to(device=device, dtype=torch.float32).contiguous()
            so0, so1 = pack(data["scales0"], data["oris0"]), pack(data["scales1"], data["oris1"])
        if (conf.depth_confidence > 0 or conf.width_confidence > 0) and m > 0 and n > 0:
            return self._forward_adaptive(kpts0, kpts1, desc0, desc1, size0, size1, so0, so1)
        d = conf.descriptor_dim
        if m > 0 and n > 0:
            self.ensure_packed(device)
            lib = nat.lib()
            s0 = torch.as_tensor(size0, device=device, dtype=torch.float32).expand(b, 2).contiguous()
            s1 = torch.as_tensor(size1, device=device, dtype=torch.float32).expand(b, 2).contiguous()
            # side-0 rows then side-1 rows: zero-copy when both views came out of ONE extractor call (adjacent
            # slices of one tensor), otherwise one concatenation each (tensor plumbing)
            kp = _pack_sides(kpts0.contiguous().float(), kpts1.contiguous().float())
            de = _pack_sides(desc0, desc1)
            so = _pack_sides(so0, so1) if so0 is not None else None
            use_graph = 0 < b * (m + n) <= int(conf.graph_max_rows or 0) and self.trace is None
            if use_graph:
                key = (b, m, n, device.index, so is not None, torch.cuda.current_stream(device).cuda_stream)
                e = self._graphs.get(key) or self._graph_entry(key, kp, de, s0, s1, so, b, m, n)
                use_graph = e["graph"] is not None
            if use_graph:
                for name, src in (("kp", kp), ("de", de), ("s0", s0), ("s1", s1), ("so", so)):
                    if src is not None:
                        e[name].copy_(src)
                e["graph"].replay()
                # the graph owns its buffers: the caller gets copies (plumbing; 6 MB at 1024 x 1024 points)
                m0, m1, ms0, ms1 = e["m0"].clone(), e["m1"].clone(), e["ms0"].clone(), e["ms1"].clone()
                scores, rows = e["scores"].clone(), e["rows"].clone()
            else:
                # every element of the outputs is written by gfc_lg_forward_packed: no fills
                m0 = torch.empty((b, m), device=device, dtype=torch.long)
                m1 = torch.empty((b, n), device=device, dtype=torch.long)
                ms0, ms1 = torch.empty((b, m), device=device), torch.empty((b, n), device=device)
                scores = torch.empty((b, m + 1, n + 1), device=device)
                # one row buffer [b*m + b*n, 256]: the library's layers work in place on it and leave the last layer's
                # descriptors there; ref_descriptors0/1 are its two halves (no copy out)
                rows = torch.empty((b * (m + n), d), device=device)
                ws = self._ws.get(lib.gfc_lg_packed_workspace_bytes(b, m, n), device)
                self._launch_packed(kp, de, s0, s1, so, b, m, n, m0, m1, ms0, ms1, scores, rows, ws, self.trace)
        else:  # the reference's early return (lightglue.py:298-303): all -1 / zeros
            m0 = torch.full((b, m), -1, device=device, dtype=torch.long)
            m1 = torch.full((b, n), -1, device=device, dtype=torch.long)
            ms0, ms1 = torch.zeros((b, m), device=device), torch.zeros((b, n), device=device)
            scores = torch.zeros((b, m + 1, n + 1), device=device)
            rows = torch.zeros((b * (m + n), d), device=device)
        ref0 = rows[: b * m].view(b, 1, m, d)
        ref1 = rows[b * m:].view(b, 1, n, d)
        # m == 0 or n == 0: the reference's early return (lightglue.py:298-303) -> all -1 / zeros
        return {
            "matches0": m0,
            "matches1": m1,
            "matching_scores0": ms0,
            "matching_scores1": ms1,
            "ref_descriptors0": ref0,
            "ref_descriptors1": ref1,
            "log_assignment": scores,
            "prune0": torch.full_like(ms0, conf.n_layers),
            "prune1": torch.full_like(ms1, conf.n_layers),
        }

    # -- several pairs of DIFFERENT sizes through one launch sequence -----------------------------
    def forward_pairs(self, items: list) -> list:
        """MI355X addition: `[self(d) for d in items]` (each `d` a batch-1 input of `forward`) as ONE matcher pass.
        The reference's evaluation loop calls the matcher pair by pair only because the images of an HPatches-style
        list differ in size (utils/export_predictions.py:36-45, datasets/hpatches.py:60); LightGlue does not care
        about image sizes once the key points exist, so pairs with their own key-point counts run together through
        gfc_lg_forward_ragged: the layers over all rows at once, the assignment head per group of equal-shape pairs.
        Same arithmetic per pair; results are returned per pair in the order given.  Pairs without key points in one
        view, adaptive depth / width and batch sizes other than 1 take the single-pair path."""
        conf = self.conf
        adaptive = conf.depth_confidence > 0 or conf.width_confidence > 0
        outs = [None] * len(items)
        rag = []
        for i, data in enumerate(items):
            for key in self.required_data_keys:
                assert key in data, f"Missing key {key} in data"
            k0, k1 = data["keypoints0"], data["keypoints1"]
            if adaptive or k0.shape[0] != 1 or k0.shape[1] == 0 or k1.shape[1] == 0 or self.training:
                outs[i] = self(data)
            else:
                rag.append(i)
        for c in range(0, len(rag), nat.GFC_LG_MAX_RAGGED_PAIRS):
            chunk = rag[c:c + nat.GFC_LG_MAX_RAGGED_PAIRS]
            for i, out in zip(chunk, self._forward_ragged([items[i] for i in chunk])):
                outs[i] = out
        return outs

    def _forward_ragged(self, items):
        conf, lib = self.conf, nat.lib()
        if not self.are_weights_initialized:
            raise RuntimeError("LightGlue weights are not loaded (conf.weights or load_state_dict)")
        device = items[0]["keypoints0"].device
        nat.require_cuda(items[0]["keypoints0"], "data['keypoints0']")
        self.ensure_packed(device)
        d, din = conf.descriptor_dim, conf.input_dim
        # equal shapes next to each other (stable): every run of equal (m, n) is one batched assignment head
        shapes = [(int(it["keypoints0"].shape[1]), int(it["keypoints1"].shape[1])) for it in items]
        order = sorted(range(len(items)), key=lambda i: shapes[i])
        kp_parts, de_parts, so_parts, s0, s1 = [], [], [], [], []
        g = 0
        while g < len(order):
            h = g
            while h < len(order) and shapes[order[h]] == shapes[order[g]]:
                h += 1
            for side in ("0", "1"):
                for i in order[g:h]:
                    it = items[i]
                    kp = it["keypoints" + side][0].float()
                    de = it["descriptors" + side][0].float()
                    assert de.shape[-1] == din
                    kp_parts.append(kp)
                    de_parts.append(de)
                    if conf.add_scale_ori:
                        sc, ori = it["scales" + side][0], it["oris" + side][0]
                        so_parts.append(torch.stack([sc.reshape(-1), ori.reshape(-1)], -1).float())
            g = h
        for i in order:
            it = items[i]
            for side, acc in (("0", s0), ("1", s1)):
                size = it.get("view" + side, {}).get("image_size")
                kp = it["keypoints" + side]
                if size is None:  # normalize_keypoints without a size: the extent of the key points (lightglue.py:31-32)
                    size = 1 + kp.float().amax(-2) - kp.float().amin(-2)
                acc.append(torch.as_tensor(size, device=device, dtype=torch.float32).reshape(-1, 2)[:1])
        b = len(order)
        kp = torch.cat(kp_parts, 0).contiguous()
        de = torch.cat(de_parts, 0).contiguous()
        so = torch.cat(so_parts, 0).contiguous() if so_parts else None
        size0, size1 = torch.cat(s0, 0).contiguous(), torch.cat(s1, 0).contiguous()
        ms = [shapes[i][0] for i in order]
        ns = [shapes[i][1] for i in order]
        cm, cn = (ctypes.c_int32 * b)(*ms), (ctypes.c_int32 * b)(*ns)
        sm, sn = sum(ms), sum(ns)
        m0 = torch.empty((sm,), device=device, dtype=torch.long)
        m1 = torch.empty((sn,), device=device, dtype=torch.long)
        sc0, sc1 = torch.empty((sm,), device=device), torch.empty((sn,), device=device)
        scores = torch.empty((sum((a + 1) * (c + 1) for a, c in zip(ms, ns)),), device=device)
        rows = torch.empty((sm + sn, d), device=device)
        ws = self._ws.get(lib.gfc_lg_ragged_workspace_bytes(b, cm, cn), device)
        nat.check(lib.gfc_lg_forward_ragged(
            ctypes.byref(self._packed[0]), nat.ptr(kp), nat.ptr(de), nat.ptr(size0), nat.ptr(size1), nat.ptr(so), b, cm,
            cn, float(conf.filter_threshold), nat.ptr(m0), nat.ptr(m1), nat.ptr(sc0), nat.ptr(sc1), nat.ptr(scores),
            nat.ptr(rows), nat.ptr(ws), ws.numel(), ctypes.byref(self.trace.c) if self.trace is not None else None,
            nat.stream_ptr(device)), "gfc_lg_forward_ragged")
        # per-pair views of the flat outputs; rows: group after group, side 0 then side 1 inside a group
        outs = [None] * b
        o0 = o1 = os_ = r = 0
        g = 0
        while g < b:
            h = g
            while h < b and (ms[h], ns[h]) == (ms[g], ns[g]):
                h += 1
            m, n, cnt = ms[g], ns[g], h - g
            for j in range(cnt):
                ms0_, ms1_ = sc0[o0:o0 + m].view(1, m), sc1[o1:o1 + n].view(1, n)
                outs[order[g + j]] = {
                    "matches0": m0[o0:o0 + m].view(1, m), "matches1": m1[o1:o1 + n].view(1, n),
                    "matching_scores0": ms0_, "matching_scores1": ms1_,
                    "ref_descriptors0": rows[r + j * m: r + (j + 1) * m].view(1, 1, m, d),
                    "ref_descriptors1": rows[r + cnt * m + j * n: r + cnt * m + (j + 1) * n].view(1, 1, n, d),
                    "log_assignment": scores[os_:os_ + (m + 1) * (n + 1)].view(1, m + 1, n + 1),
                    "prune0": torch.full_like(ms0_, conf.n_layers), "prune1": torch.full_like(ms1_, conf.n_layers),
                }
                o0, o1, os_ = o0 + m, o1 + n, os_ + (m + 1) * (n + 1)
            r += cnt * (m + n)
            g = h
        return outs

    # -- adaptive depth / width (lightglue.py:500-521,555-580) -----------------------------------
    def _forward_adaptive(self, kpts0, kpts1, desc0, desc1, size0, size1, so0=None, so1=None):
        """Early stopping (`depth_confidence`) and point pruning (`width_confidence`); batch size 1 like the
        reference (`assert b == 1`, lightglue.py:501,507).  The host drives `gfc_lg_layer` layer by layer, takes
        the stop / prune decisions on the token confidences and matchabilities computed by `gfc_lg_rowdot`
        (one small device->host read per layer, as `check_if_stop` does in the reference) and re-packs the
        surviving rows between layers (index_select: plumbing)."""
        conf, lib = self.conf, nat.lib()
        b, m, _ = kpts0.shape
        n = kpts1.shape[1]
        assert b == 1
        device = kpts0.device
        params = self.ensure_packed(device)[0]
        st = nat.stream_ptr(device)
        d = conf.descriptor_dim
        do_early_stop, do_prune = conf.depth_confidence > 0, conf.width_confidence > 0
        # packed rows: image 0 first
        kp = torch.cat([kpts0[0], kpts1[0]], 0).float().contiguous()
        x = torch.empty((m + n, d), device=device, dtype=torch.float32)
        if conf.input_dim == d:
            x[:m], x[m:] = desc0[0], desc1[0]
        else:
            din = conf.input_dim
            xin = torch.cat([desc0[0], desc1[0]], 0).contiguous()
            nat.check(lib.gfc_linear(nat.ptr(xin), din, din, None, 0, 0, params.input_proj_w, din, params.input_proj_b,
                                     None, None, 1.0, None, None, None, 0, nat.ptr(x), d, m + n, d, st), "input_proj")
        sizes = torch.stack([torch.as_tensor(size0, device=device, dtype=torch.float32).reshape(-1, 2)[0],
                             torch.as_tensor(size1, device=device, dtype=torch.float32).reshape(-1, 2)[0]]).contiguous()
        row0 = torch.tensor([0, m], dtype=torch.int32, device=device)
        cnt = torch.tensor([m, n], dtype=torch.int32, device=device)
        cos = torch.empty((m + n, 64), device=device)
        sin = torch.empty((m + n, 64), device=device)
        so = torch.cat([so0[0], so1[0]], 0).contiguous() if so0 is not None else None
        nat.check(lib.gfc_lg_posenc(nat.ptr(kp), nat.ptr(so), nat.ptr(sizes), nat.ptr(row0), nat.ptr(cnt), 2, max(m, n),
                                    params.posenc_wr, 4 if so is not None else 2, nat.ptr(cos), nat.ptr(sin), st),
                  "gfc_lg_posenc")
        ind0 = torch.arange(m, device=device)
        ind1 = torch.arange(n, device=device)
        prune0 = torch.ones((1, m), device=device, dtype=torch.long)
        prune1 = torch.ones((1, n), device=device, dtype=torch.long)
        thresholds = self.confidence_thresholds.tolist()
        cm, cn = m, n
        last = conf.n_layers - 1
        for i in range(conf.n_layers):
            self_p = torch.tensor([[0, cm, 0, cm], [cm, cn, cm, cn]], dtype=torch.int32, device=device)
            cross_p = torch.tensor([[0, cm, cm, cn], [cm, cn, 0, cm]], dtype=torch.int32, device=device)
            ws = self._ws.get(lib.gfc_lg_layer_workspace_bytes(cm + cn), device)
            nat.check(lib.gfc_lg_layer(ctypes.byref(params), i, nat.ptr(x), nat.ptr(cos), nat.ptr(sin), cm + cn,
                                       nat.ptr(self_p), nat.ptr(cross_p), 2, max(cm, cn), nat.ptr(ws), ws.numel(), st),
                      "gfc_lg_layer")
            last = i
            if i == conf.n_layers - 1:
                break
            tok = None
            if do_early_stop:
                tok = torch.empty((cm + cn,), device=device)
                nat.check(lib.gfc_lg_rowdot(nat.ptr(x), d, cm + cn, params.token_w[i], params.token_b[i], 1,
                                            nat.ptr(tok), st), "gfc_lg_rowdot")
                # check_if_stop (lightglue.py:569-580): the ratio is taken over the ORIGINAL m + n points
                ratio = 1.0 - (tok < thresholds[i]).float().sum() / (m + n)
                if ratio.item() > conf.depth_confidence:
                    break
            if do_prune:
                sc = torch.empty((cm + cn,), device=device)
                nat.check(lib.gfc_lg_rowdot(nat.ptr(x), d, cm + cn, params.matchability_w[i],
                                            params.matchability_b[i], 1, nat.ptr(sc), st), "gfc_lg_rowdot")
                keep = sc > (1 - conf.width_confidence)  # get_pruning_mask, lightglue.py:560-567
                if tok is not None:
                    keep = keep | (tok <= thresholds[i])
                keep0 = torch.where(keep[:cm])[0]
                keep1 = torch.where(keep[cm:])[0]
                ind0, ind1 = ind0[keep0], ind1[keep1]
                rows = torch.cat([keep0, keep1 + cm])
                x, cos, sin = x[rows].contiguous(), cos[rows].contiguous(), sin[rows].contiguous()
                prune0[:, ind0] += 1
                prune1[:, ind1] += 1
                cm, cn = int(keep0.numel()), int(keep1.numel())
                if cm == 0 or cn == 0:
                    break
        m0 = torch.full((1, m), -1, device=device, dtype=torch.long)
        m1 = torch.full((1, n), -1, device=device, dtype=torch.long)
        ms0 = torch.zeros((1, m), device=device)
        ms1 = torch.zeros((1, n), device=device)
        scores = torch.zeros((1, cm + 1, cn + 1), device=device)
        if cm > 0 and cn > 0:
            pm0 = torch.empty((1, cm), device=device, dtype=torch.long)
            pm1 = torch.empty((1, cn), device=device, dtype=torch.long)
            ps0, ps1 = torch.empty((1, cm), device=device), torch.empty((1, cn), device=device)
            ws = self._ws.get(lib.gfc_lg_assign_workspace_bytes(1, cm, cn), device)
            x1 = x[cm:]
            nat.check(lib.gfc_lg_assign(ctypes.byref(params), last, nat.ptr(x), ctypes.c_void_p(x1.data_ptr()), 1, cm,
                                        cn, float(conf.filter_threshold), nat.ptr(pm0), nat.ptr(pm1), nat.ptr(ps0),
                                        nat.ptr(ps1), nat.ptr(scores), nat.ptr(ws), ws.numel(), st), "gfc_lg_assign")
            if do_prune:  # scatter back to the un-pruned indexing (lightglue.py:527-536)
                m0[:, ind0] = torch.where(pm0 == -1, -1, ind1[pm0.clamp(min=0)])
                m1[:, ind1] = torch.where(pm1 == -1, -1, ind0[pm1.clamp(min=0)])
                ms0[:, ind0], ms1[:, ind1] = ps0, ps1
            else:
                m0, m1, ms0, ms1 = pm0, pm1, ps0, ps1
        if not do_prune:
            prune0 = torch.ones_like(ms0) * conf.n_layers
            prune1 = torch.ones_like(ms1) * conf.n_layers
        return {
            "matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1,
            "ref_descriptors0": x[None, None, :cm], "ref_descriptors1": x[None, None, cm:],
            "log_assignment": scores, "prune0": prune0, "prune1": prune1,
            "stop_layer": torch.full((1,), last + 1, device=device, dtype=torch.long),
        }

    def loss(self, pred, data):
        raise NotImplementedError("training loss (lightglue.py:588-637) is out of scope")


__main_model__ = LightGlue

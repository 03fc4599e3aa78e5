"""Randomised shape sweep of gfc_conv3x3_wino / gfc_lg_assign / gfc_disk_nms_select / gfc_linear / gfc_attention /
gfc_ffn_fused / gfc_lg_forward_ragged / gfc_sp_detector_head / gfc_preprocess_resize against torch float64 / the oracle / the single-pair calls
(run on the GPU box: python tools/micro/fuzz_shapes.py [n_cases]).  Prints the worst error and any failing shape."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import _native as nat  # noqa: E402
from oracle import disk as odisk  # noqa: E402
from oracle import lightglue as olg  # noqa: E402

def run(n_cases=60, seed=2024):
    """-> list of failing cases (empty when every shape agrees)."""
    lib = nat.lib()
    DEV = torch.device("cuda", 0)
    st = nat.stream_ptr(DEV)
    g = torch.Generator().manual_seed(seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    worst, bad = 0.0, []
    for case in range(n_cases):
        cin, cout = [16, 32, 64, 128][ri(0, 3)], [64, 128][ri(0, 1)]
        h, w, b, pool = ri(1, 70), ri(1, 70), ri(1, 3), ri(0, 1)
        if pool and (h < 2 or w < 2):
            continue
        x = torch.randn((b, cin, h, w), generator=g)
        wt = torch.randn((cout, cin, 3, 3), generator=g) / (3 * cin ** 0.5)
        bias = torch.randn((cout,), generator=g) * 0.1
        ref = F.relu(F.conv2d(x.double(), wt.double(), bias.double(), padding=1))
        if pool:
            ref = F.max_pool2d(ref, 2, 2)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        wd, bd = wt.to(DEV), bias.to(DEV)
        ww = torch.empty((16 * cout * cin,), device=DEV)
        nat.check(lib.gfc_pack_conv3x3_wino(nat.ptr(wd), nat.ptr(ww), cout, cin, st), "pack")
        ho, wo = (h // 2, w // 2) if pool else (h, w)
        y = torch.full((b, ho, wo, cout), float("nan"), device=DEV)
        nat.check(lib.gfc_conv3x3_wino(nat.ptr(xd), nat.ptr(ww), nat.ptr(bd), None, None, nat.ptr(y), b, h, w, cin, cout, 1,
                                       pool, st), "wino")
        torch.cuda.synchronize()
        err = (y.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item() if ref.numel() else 0.0
        worst = max(worst, err)
        if not err < 2e-5:
            bad.append(("wino", cin, cout, h, w, b, pool, err))
    print("winograd: worst", worst, "bad", bad)
    failures = list(bad)

    # ---- the fused stems (conv1a + conv1b + pool): F(2x2,3x3) and F(4x4,3x3), any H x W, a guard band behind the output ----
    bad, worst = [], 0.0
    for case in range(n_cases):
        h, w, b, bn = ri(2, 150), ri(2, 150), ri(1, 4), ri(0, 1)
        if case % 7 == 0:
            b = ri(10, 40)  # more items than workgroups of the persistent F(4,3) launch at small sizes
            h, w = ri(2, 40), ri(2, 70)
        img = torch.rand((b, 1, h, w), generator=g)
        w1 = torch.randn((64, 1, 3, 3), generator=g) / 3
        b1 = torch.randn((64,), generator=g) * 0.1
        w2 = torch.randn((64, 64, 3, 3), generator=g) / 24
        b2 = torch.randn((64,), generator=g) * 0.1
        s1 = s2 = t1 = t2 = None
        ref = F.relu(F.conv2d(img.double(), w1.double(), b1.double(), padding=1))
        if bn:
            s1, t1 = torch.rand((64,), generator=g) + 0.5, torch.randn((64,), generator=g) * 0.1
            s2, t2 = torch.rand((64,), generator=g) + 0.5, torch.randn((64,), generator=g) * 0.1
            s2[::3] *= -1
            ref = ref * s1.double()[None, :, None, None] + t1.double()[None, :, None, None]
        ref = F.relu(F.conv2d(ref, w2.double(), b2.double(), padding=1))
        if bn:
            ref = ref * s2.double()[None, :, None, None] + t2.double()[None, :, None, None]
        ref = F.max_pool2d(ref, 2, 2)
        dd = lambda t: None if t is None else t.to(DEV).contiguous()  # noqa: E731
        keep = [dd(t) for t in (img.reshape(b, h, w), w1.reshape(64, 9).t(), b1, s1, t1, w2, b2, s2, t2)]
        n_out = b * (h // 2) * (w // 2) * 64
        for variant in ("f23", "f43"):
            if variant == "f23":
                w2w = torch.empty((16 * 64 * 64,), device=DEV)
                nat.check(lib.gfc_pack_conv3x3_wino(nat.ptr(keep[5]), nat.ptr(w2w), 64, 64, st), "pack")
                fn = lib.gfc_sp_stem_wino
            else:
                w2w = torch.empty((36 * 64 * 64,), device=DEV)
                nat.check(lib.gfc_pack_conv3x3_wino43(nat.ptr(keep[5]), nat.ptr(w2w), 64, 64, st), "pack43")
                fn = lib.gfc_sp_stem_wino43
            buf = torch.full((n_out + 4096,), float("nan"), device=DEV)
            nat.check(fn(nat.ptr(keep[0]), nat.ptr(keep[1]), nat.ptr(keep[2]), nat.ptr(keep[3]), nat.ptr(keep[4]), nat.ptr(w2w),
                         nat.ptr(keep[6]), nat.ptr(keep[7]), nat.ptr(keep[8]), nat.ptr(buf), b, h, w, st), "stem")
            torch.cuda.synchronize()
            y = buf[:n_out].view(b, h // 2, w // 2, 64)
            err = (y.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item() if n_out else 0.0
            guard_clean = bool(torch.isnan(buf[n_out:]).all())
            worst = max(worst, err if err == err else 0.0)
            if not (err < 4e-5 and guard_clean):
                bad.append(("stem", variant, h, w, b, bn, err, guard_clean))
    print("stems: worst", worst, "bad", bad)
    failures += bad

    # ---- DISK's 5x5 convolution ([InstanceNorm -> PReLU ->] conv, channel-sliced output, channel sub-ranges) ----
    bad, worst = [], 0.0
    for case in range(max(n_cases // 2, 8)):
        cin, cout = 4 * ri(1, 34), ri(1, 140)
        h, w, b, gated = ri(1, 50), ri(1, 50), ri(1, 3), ri(0, 1)
        if 480 % (cin // 4):
            gated = 0  # the statistics kernel covers C/4 dividing 480 (every channel count of the network)
        if gated and h * w < 2:
            continue
        x = torch.randn((b, cin, h, w), generator=g) * 2 + 0.3
        wt = torch.randn((cout, cin, 5, 5), generator=g) / (5 * cin ** 0.5)
        bias = torch.randn((cout,), generator=g) * 0.1
        slope = torch.rand((cin,), generator=g) * 0.5
        ref = x.double()
        if gated:
            ref = F.prelu(F.instance_norm(ref, eps=1e-5), slope.double())
        ref = F.conv2d(ref, wt.double(), bias.double(), padding=2)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        wp = torch.empty((lib.gfc_disk_conv5x5_packed_floats(cout, cin),), device=DEV)
        wdev, bd = wt.to(DEV), bias.to(DEV)
        nat.check(lib.gfc_disk_pack_conv5x5(nat.ptr(wdev), nat.ptr(wp), cout, cin, st), "pack5")
        mean = rstd = sl = None
        if gated:
            mean, rstd, sl = torch.empty((b, cin), device=DEV), torch.empty((b, cin), device=DEV), slope.to(DEV)
            ws = torch.empty(lib.gfc_disk_instnorm_workspace_bytes(b, cin), dtype=torch.uint8, device=DEV)
            nat.check(lib.gfc_disk_instnorm_stats(nat.ptr(xd), b, h, w, cin, 1e-5, nat.ptr(mean), nat.ptr(rstd), nat.ptr(ws),
                                                  ws.numel(), st), "stats")
        first = 32 * ri(0, (cout - 1) // 32)
        count = ri(1, cout - first)
        pad = 4 * ri(0, 3)
        y = torch.full((b, h, w, count + pad), float("nan"), device=DEV)
        nat.check(lib.gfc_disk_conv5x5(nat.ptr(xd), nat.ptr(mean), nat.ptr(rstd), nat.ptr(sl), nat.ptr(wp), nat.ptr(bd),
                                       nat.ptr(y), count + pad, b, h, w, cin, cout, first, count, st), "conv5x5")
        torch.cuda.synchronize()
        r = ref[:, first:first + count]
        err = ((y[..., :count].permute(0, 3, 1, 2).double().cpu() - r).abs() / (1 + r.abs())).max().item()
        clean = bool(torch.isnan(y[..., count:]).all())
        worst = max(worst, err if err == err else 0.0)
        if not (err < 3e-5 and clean):
            bad.append(("conv5x5", cin, cout, h, w, b, gated, first, count, err, clean))
    print("conv5x5: worst", worst, "bad", bad)
    failures += bad

    bad = []
    for case in range(n_cases):
        h, w, b = ri(1, 90), ri(1, 90), ri(1, 3)
        n = [None, ri(1, 50), ri(50, 4000)][ri(0, 2)]
        window = [1, 3, 5, 7, 9][ri(0, 4)]
        heat = torch.randn((b, 1, h, w), generator=g)
        if ri(0, 1):
            heat = (heat * 3).round() / 3
        ref = odisk.heatmap_to_keypoints(heat, n, window, 0.0)
        cap = n if n is not None else h * w
        kp = torch.full((b, cap, 2), -7.0, device=DEV)
        sc = torch.full((b, cap), -7.0, device=DEV)
        cnt = torch.empty((b,), dtype=torch.int32, device=DEV)
        ws = torch.empty(lib.gfc_disk_select_workspace_bytes(b, h, w), dtype=torch.uint8, device=DEV)
        hd = heat.reshape(b, h, w).contiguous().to(DEV)
        nat.check(lib.gfc_disk_nms_select(nat.ptr(hd), b, h, w, window, 0.0, -1 if n is None else n, cap, nat.ptr(kp), nat.ptr(sc),
                                          nat.ptr(cnt), nat.ptr(ws), ws.numel(), st), "disk")
        torch.cuda.synchronize()
        for i, (xy, s) in enumerate(ref):
            c = int(cnt[i])
            if c != xy.shape[0] or not torch.equal(kp[i, :c].cpu(), xy.float()) or not torch.equal(sc[i, :c].cpu(), s):
                bad.append(("disk", h, w, b, n, window, i, c, xy.shape[0]))
    print("disk select: bad", bad)
    failures += bad

    bad, worst = [], 0.0
    for case in range(max(n_cases // 3, 8)):
        b, m, n = ri(1, 3), ri(1, 1300), ri(1, 1300)
        x0, x1 = torch.randn((b, m, 256), generator=g), torch.randn((b, n, 256), generator=g)
        sd = {"log_assignment.0.final_proj.weight": torch.randn((256, 256), generator=g) / 8,
              "log_assignment.0.final_proj.bias": torch.randn((256,), generator=g) * 0.1,
              "log_assignment.0.matchability.weight": torch.randn((1, 256), generator=g) / 16,
              "log_assignment.0.matchability.bias": torch.randn((1,), generator=g)}
        ref = olg.match_assignment(sd, "log_assignment.0", x0, x1)
        r0, r1, _, _ = olg.filter_matches(ref, 0.1)
        keep = [sd[k].to(DEV).contiguous() for k in sd]
        p = nat.LgParams()
        p.n_layers, p.input_dim = 1, 256
        p.final_proj_w[0], p.final_proj_b[0] = keep[0].data_ptr(), keep[1].data_ptr()
        p.matchability_w[0], p.matchability_b[0] = keep[2].reshape(-1).data_ptr(), keep[3].data_ptr()
        m0 = torch.empty((b, m), dtype=torch.long, device=DEV)
        m1 = torch.empty((b, n), dtype=torch.long, device=DEV)
        s0, s1 = torch.empty((b, m), device=DEV), torch.empty((b, n), device=DEV)
        la = torch.full((b, m + 1, n + 1), float("nan"), device=DEV)
        ws = torch.full((lib.gfc_lg_assign_workspace_bytes(b, m, n),), 0xFF, dtype=torch.uint8, device=DEV)
        x0d, x1d = x0.reshape(b * m, 256).to(DEV), x1.reshape(b * n, 256).to(DEV)
        nat.check(lib.gfc_lg_assign(ctypes.byref(p), 0, nat.ptr(x0d), nat.ptr(x1d), b, m, n, 0.1, nat.ptr(m0), nat.ptr(m1),
                                    nat.ptr(s0), nat.ptr(s1), nat.ptr(la), nat.ptr(ws), ws.numel(), st), "assign")
        torch.cuda.synchronize()
        err = ((la.cpu() - ref).abs() / (1 + ref.abs())).max().item()
        worst = max(worst, err)
        if not (err < 1e-4 and torch.equal(m0.cpu(), r0) and torch.equal(m1.cpu(), r1)):
            bad.append(("assign", b, m, n, err, int((m0.cpu() != r0).sum()), int((m1.cpu() != r1).sum())))
    print("assign: worst", worst, "bad", bad)
    failures += bad

    # ---- gfc_linear: ragged M / N, both K sources, every epilogue (bias, affine, alpha, rotary, residual) ----
    bad, worst = [], 0.0
    for case in range(n_cases):
        m, n = ri(1, 900), ri(1, 12) * 64 if ri(0, 1) else ri(1, 700)
        k0, k1 = ri(1, 8) * 32, ri(0, 1) * ri(1, 4) * 32
        mode = ri(0, 3)  # 0 plain, 1 affine + alpha, 2 rotary (first rot_cols columns), 3 residual
        if mode == 2:
            n = ri(1, 12) * 64
        a0 = torch.randn((m, k0), generator=g)
        a1 = torch.randn((m, k1), generator=g) if k1 else None
        w = torch.randn((n, k0 + k1), generator=g) / (k0 + k1) ** 0.5
        bias = torch.randn((n,), generator=g) * 0.1 if ri(0, 3) else None
        ref = torch.cat([a0, a1], 1).double() if k1 else a0.double()
        ref = ref @ w.double().t()
        if bias is not None:
            ref = ref + bias.double()
        scale = shift = res = cos = sin = None
        alpha, rot_cols = 1.0, 0
        if mode == 1:
            scale, shift, alpha = torch.rand((n,), generator=g) + 0.5, torch.randn((n,), generator=g) * 0.1, 0.25
            ref = (ref * scale.double() + shift.double()) * alpha
        elif mode == 2:
            rot_cols = ri(1, n // 64) * 64
            ang = torch.randn((m, 32), generator=g) * 2
            cos, sin = ang.cos().repeat_interleave(2, -1), ang.sin().repeat_interleave(2, -1)
            t = ref[:, :rot_cols].reshape(m, rot_cols // 64, 64)
            rot = torch.stack([-t[..., 1::2], t[..., 0::2]], -1).reshape(t.shape)
            ref = torch.cat([(t * cos.double()[:, None] + rot * sin.double()[:, None]).reshape(m, rot_cols), ref[:, rot_cols:]], 1)
        elif mode == 3:
            res = torch.randn((m, n), generator=g)
            ref = ref + res.double()
        dd = lambda t: None if t is None else t.to(DEV).contiguous()  # noqa: E731
        keep = [dd(t) for t in (a0, a1, w, bias, scale, shift, res, cos, sin)]
        y = torch.full((m, n), float("nan"), device=DEV)
        rc = lib.gfc_linear(nat.ptr(keep[0]), k0, k0, nat.ptr(keep[1]), k1, k1, nat.ptr(keep[2]), k0 + k1, nat.ptr(keep[3]),
                            nat.ptr(keep[4]), nat.ptr(keep[5]), alpha, nat.ptr(keep[6]), nat.ptr(keep[7]), nat.ptr(keep[8]),
                            rot_cols, nat.ptr(y), n, m, n, st)
        torch.cuda.synchronize()
        if n % 4 and mode == 3:  # unaligned residual rows: allowed to be refused
            if rc != 0:
                continue
        err = float("nan") if rc != 0 else ((y.double().cpu() - ref).abs() / (1 + ref.abs())).max().item()
        worst = max(worst, err if err == err else 0.0)
        if not err < 3e-5:
            bad.append(("linear", m, n, k0, k1, mode, rot_cols, rc, err))
    print("linear: worst", worst, "bad", bad)
    failures += bad

    # ---- gfc_attention: ragged problem sets, with and without the key-split scratch ----
    bad, worst = [], 0.0
    for case in range(max(n_cases // 4, 8)):
        shapes = [(ri(1, 600), ri(1, 600)) for _ in range(ri(1, 4))]
        rows = sum(max(a, b) for a, b in shapes)
        q, k, v = (torch.randn((rows, 256), generator=g) * s for s in (1.5, 1.5, 1.0))
        probs, r0 = [], 0
        for nq, nk in shapes:
            probs.append([r0, nq, r0, nk])
            r0 += max(nq, nk)
        qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
        o = torch.full((rows, 256), float("nan"), device=DEV)
        pt = torch.tensor(probs, dtype=torch.int32, device=DEV)
        max_nq = max(a for a, _ in shapes)
        ws = None
        if ri(0, 1):
            nb = lib.gfc_attention_workspace_bytes(len(shapes), max_nq, 4)
            ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=DEV) if nb else None
        nat.check(lib.gfc_attention(nat.ptr(qd), 256, nat.ptr(kd), 256, nat.ptr(vd), 256, nat.ptr(o), 256, nat.ptr(pt),
                                    len(shapes), max_nq, 4, 0.125, nat.ptr(ws), 0 if ws is None else ws.numel(), st), "attention")
        torch.cuda.synchronize()
        oc = o.cpu()
        for r, nq, _, nk in probs:
            qq = q[r:r + nq].double().view(nq, 4, 64).transpose(0, 1)
            kk = k[r:r + nk].double().view(nk, 4, 64).transpose(0, 1)
            vv = v[r:r + nk].double().view(nk, 4, 64).transpose(0, 1)
            ref = (torch.softmax(qq @ kk.transpose(1, 2) * 0.125, -1) @ vv).transpose(0, 1).reshape(nq, 256)
            err = (oc[r:r + nq].double() - ref).abs().max().item()
            worst = max(worst, err if err == err else 0.0)
            if not err < 2e-5:
                bad.append(("attention", shapes, (nq, nk), err))
    print("attention: worst", worst, "bad", bad)
    failures += bad

    # ---- gfc_ffn_fused (whole FFN + residual in one kernel) at any row count, against float64 and the two-kernel path ----
    bad, worst = [], 0.0
    for case in range(max(n_cases // 6, 6)):
        m = [ri(1, 300), ri(300, 5000), 128 * ri(1, 40), 16384 + ri(0, 300)][case % 4]
        x, msg = torch.randn((m, 256), generator=g), torch.randn((m, 256), generator=g) * 1.5
        w0, b0 = torch.randn((512, 512), generator=g) / 512 ** 0.5, torch.randn((512,), generator=g)
        ga, be = torch.rand((512,), generator=g) + 0.5, torch.randn((512,), generator=g) * 0.2
        w3, b3 = torch.randn((256, 512), generator=g) / 512 ** 0.5, torch.randn((256,), generator=g)
        dv = [t.to(DEV).contiguous() for t in (x, msg, w0, b0, ga, be, w3, b3)]
        y = torch.full((m, 256), float("nan"), device=DEV)
        nat.check(lib.gfc_ffn_fused(nat.ptr(dv[0]), 256, 256, nat.ptr(dv[1]), 256, 256, nat.ptr(dv[2]), 512, nat.ptr(dv[3]),
                                    nat.ptr(dv[4]), nat.ptr(dv[5]), nat.ptr(dv[6]), 512, nat.ptr(dv[7]), nat.ptr(dv[0]),
                                    nat.ptr(y), 256, m, st), "ffn_fused")
        hb = torch.empty((m, 512), device=DEV)
        nat.check(lib.gfc_linear_layernorm_gelu(nat.ptr(dv[0]), 256, 256, nat.ptr(dv[1]), 256, 256, nat.ptr(dv[2]), 512,
                                                nat.ptr(dv[3]), nat.ptr(dv[4]), nat.ptr(dv[5]), nat.ptr(hb), 512, m, 512, st), "ln")
        y2 = torch.empty((m, 256), device=DEV)
        nat.check(lib.gfc_linear(nat.ptr(hb), 512, 512, None, 0, 0, nat.ptr(dv[6]), 512, nat.ptr(dv[7]), None, None, 1.0,
                                 nat.ptr(dv[0]), None, None, 0, nat.ptr(y2), 256, m, 256, st), "ffn3")
        torch.cuda.synchronize()
        xd, md = x.double(), msg.double()
        ref = xd + F.linear(F.gelu(F.layer_norm(F.linear(torch.cat([xd, md], 1), w0.double(), b0.double()), (512,),
                                                ga.double(), be.double(), 1e-5)), w3.double(), b3.double())
        err = (y.double().cpu() - ref).abs().max().item()
        worst = max(worst, err if err == err else 0.0)
        if not err < 3e-5 or not torch.equal(y, y2):
            bad.append(("ffn_fused", m, err, bool(torch.equal(y, y2))))
    print("ffn_fused: worst", worst, "bad", bad)
    failures += bad

    # ---- gfc_lg_forward_ragged: random sets of pairs with their own key-point counts against single-pair calls ----
    from glue_factory_colon_amd import lightglue

    bad, worst = [], 0.0
    mods = {d: lightglue.LightGlue({"weights": "synthetic", "filter_threshold": 0.1, "input_dim": d,
                                    "n_layers": 3}).eval().to(DEV) for d in (256, 128)}
    for case in range(max(n_cases // 8, 5)):
        dim = 256 if case % 3 else 128
        mdl = mods[dim]
        nb = ri(2, 7)
        items = []
        for _ in range(nb):
            m_, n_ = (ri(1, 700), ri(1, 700)) if ri(0, 3) else (320, 320)  # some equal shapes: batched assignment groups
            kp0, kp1 = torch.rand((1, m_, 2), generator=g) * 600, torch.rand((1, n_, 2), generator=g) * 600
            d0 = F.normalize(torch.randn((1, m_, dim), generator=g), dim=-1)
            d1 = F.normalize(torch.randn((1, n_, dim), generator=g), dim=-1)
            size = torch.tensor([[640.0, 480.0]], device=DEV)
            items.append({"keypoints0": kp0.to(DEV), "keypoints1": kp1.to(DEV), "descriptors0": d0.to(DEV),
                          "descriptors1": d1.to(DEV), "view0": {"image_size": size}, "view1": {"image_size": size}})
        with torch.no_grad():
            single = [mdl(it) for it in items]
            multi = mdl.forward_pairs(items)
        torch.cuda.synchronize()
        for i, (a, b_) in enumerate(zip(single, multi)):
            same = all(torch.equal(a[k], b_[k]) for k in ("matches0", "matches1"))
            err = max((a[k] - b_[k]).abs().max().item() if a[k].numel() else 0.0
                      for k in ("matching_scores0", "matching_scores1", "ref_descriptors0", "ref_descriptors1"))
            la = ((a["log_assignment"] - b_["log_assignment"]).abs() / (1 + a["log_assignment"].abs())).max().item()
            worst = max(worst, err, la)
            if not same or not err < 1e-4 or not la < 1e-4:
                bad.append(("ragged", dim, [tuple(it["keypoints0"].shape[1:2]) + tuple(it["keypoints1"].shape[1:2]) for it in items],
                            i, same, err, la))
    print("ragged forward: worst", worst, "bad", bad)
    failures += bad

    # ---- round 5: fused detector head (any cell count, BN or not, row pitch 256 or 512) against float64 ----
    bad, worst = [], 0.0
    for case in range(n_cases):
        b_, h8, w8, bn, lda = ri(1, 4), ri(1, 40), ri(1, 50), ri(0, 1), (256, 512)[ri(0, 1)]
        rows = b_ * h8 * w8
        hidden = torch.randn((rows, lda), generator=g).clamp_(min=0)
        wt = torch.randn((65, 256), generator=g) / 8
        bias = torch.randn((65,), generator=g)
        scale = (torch.rand((65,), generator=g) + 0.5) if bn else None
        shift = torch.randn((65,), generator=g) if bn else None
        hd, wd, bd = hidden.to(DEV), wt.to(DEV), bias.to(DEV)
        scd, shd = (scale.to(DEV), shift.to(DEV)) if bn else (None, None)
        guard = 64
        heat = torch.full((b_ * h8 * 8 * w8 * 8 + guard,), float("nan"), device=DEV)
        nat.check(lib.gfc_sp_detector_head(nat.ptr(hd), lda, nat.ptr(wd), nat.ptr(bd), nat.ptr(scd), nat.ptr(shd), b_, h8, w8,
                                           nat.ptr(heat), st), "gfc_sp_detector_head")
        torch.cuda.synchronize()
        logits = hidden[:, :256].double() @ wt.double().T + bias.double()
        if bn:
            logits = logits * scale.double() + shift.double()
        ref = torch.softmax(logits, 1)[:, :64].reshape(b_, h8, w8, 8, 8).permute(0, 1, 3, 2, 4).reshape(-1)
        err = (heat[:-guard].double().cpu() - ref).abs().max().item()
        untouched = bool(torch.isnan(heat[-guard:]).all())
        worst = max(worst, err)
        if not err < 2e-6 or not untouched:
            bad.append(("det_head", b_, h8, w8, bn, lda, err, untouched))
    print("detector head: worst", worst, "bad", bad)
    failures += bad

    # ---- round 5: resize kernel (uint8 HWC / float CHW, any sizes up and down, antialias on / off) against the oracle ----
    from glue_factory_colon_amd import image_preprocessor as ip
    from oracle import preprocess as opp
    bad, worst = [], 0.0
    for case in range(n_cases):
        h, w, c = ri(8, 260), ri(8, 260), (1, 3)[ri(0, 1)]
        oh, ow = ri(4, 200), ri(4, 200)
        if max(h / oh, w / ow) > 12:  # (kernel windows above 63 taps are refused: > ~30x down-scaling)
            continue
        ac, aa = (None, True)[ri(0, 1)] if ri(0, 3) == 0 else None, bool(ri(0, 1))
        u8 = torch.randint(0, 256, (h, w, c), generator=g, dtype=torch.uint8)
        x = opp.numpy_image_to_torch(u8.numpy() if c == 3 else u8[..., 0].numpy())
        ref = opp.kornia_resize(x, (oh, ow), ac, aa)
        out_u = ip.resize((u8 if c == 3 else u8[..., 0].contiguous()).to(DEV), (oh, ow), ac, aa).cpu()
        out_f = ip.resize(x.to(DEV), (oh, ow), ac, aa).cpu()
        err = (out_u - ref).abs().max().item()
        same = torch.equal(out_u, out_f)
        worst = max(worst, err)
        if not err < 3e-6 or not same:
            bad.append(("resize", h, w, c, oh, ow, ac, aa, err, same))
    print("resize: worst", worst, "bad", bad)
    return failures + bad

if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 2024) else 0)

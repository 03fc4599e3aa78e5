"""Decision experiment (round 3): Winograd F(4x4,3x3) in fp32 through the whole SuperPoint-open stack,
CPU emulation (filters transformed in float64 and rounded once, data / output transforms and products in fp32, the
same arithmetic a HIP kernel would perform up to accumulation order).

For every test image: heat-map / descriptor error against a float64 evaluation for direct fp32, F(2x2,3x3) and
F(4x4,3x3) (all 3x3 layers after conv1a, or conv1b only), and the key-point flips among the top-k against the direct fp32
stack (what the oracle computes).  Build criterion: 0 flips on every case AND heat-map error < 1e-5.

    python tools/micro/winograd_f43_numerics.py            # 6 VGA images (bench / test seeds) + 1 image 1024 x 1024
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glue_factory_colon_amd import synthetic, weights  # noqa: E402
from oracle import superpoint as osp  # noqa: E402

torch.set_num_threads(8)
sd = weights.superpoint_open_state_dict(0)

D = torch.float64
W2 = dict(BT=torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=D),
          G=torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=D),
          AT=torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=D), m=2)
W4 = dict(BT=torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                           [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=D),
          G=torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                          [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=D),
          AT=torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=D),
          m=4)


def wino_conv(x, w, b, T):
    """x [B,C,H,W] fp32, 3x3 pad 1, F(m x m, 3x3) emulated in fp32."""
    m = T["m"]
    a = m + 2
    Bn, C, H, W = x.shape
    Co = w.shape[0]
    U = (T["G"] @ w.double() @ T["G"].T).float()  # [Co,C,a,a]  (pack time, float64, rounded once)
    Hp, Wp = (H + m - 1) // m * m, (W + m - 1) // m * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    t = xp.unfold(2, a, m).unfold(3, a, m)  # [B,C,th,tw,a,a]
    bt, at = T["BT"].float(), T["AT"].float()
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", bt, t, bt)
    out = x.new_zeros((Bn, Co, t.shape[2], t.shape[3], a, a))
    for c0 in range(0, Co, 16):  # bounded temporaries
        out[:, c0:c0 + 16] = torch.einsum("ocil,bcyxil->boyxil", U[c0:c0 + 16], V)
    Y = torch.einsum("ij,boyxjk,lk->boyxil", at, out, at)  # [B,Co,th,tw,m,m]
    th, tw = Y.shape[2], Y.shape[3]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, Co, th * m, tw * m)[:, :, :H, :W]
    return Y + b.view(1, -1, 1, 1)


def direct(x, w, b):
    return F.conv2d(x, w, b, padding=1)


def dense(image, conv_for, dt=torch.float32):
    """conv_for(layer name) -> convolution function for that 3x3 layer."""
    x = image.to(dt)

    def block(x, prefix, relu=True):
        w = sd[prefix + ".conv.weight"].to(dt)
        bb = sd[prefix + ".conv.bias"].to(dt)
        x = conv_for(prefix)(x, w, bb) if (w.shape[-1] == 3 and w.shape[1] > 1) else F.conv2d(x, w, bb, padding=w.shape[-1] // 2)
        if relu:
            x = F.relu(x)
        return F.batch_norm(x, sd[prefix + ".bn.running_mean"].to(dt), sd[prefix + ".bn.running_var"].to(dt),
                            sd[prefix + ".bn.weight"].to(dt), sd[prefix + ".bn.bias"].to(dt), training=False, eps=0.001)

    for bk in range(4):
        x = block(x, f"backbone.{bk}.0")
        x = block(x, f"backbone.{bk}.1")
        if bk < 3:
            x = F.max_pool2d(x, 2, 2)
    desc = block(block(x, "descriptor.0"), "descriptor.1", relu=False)
    logits = block(block(x, "detector.0"), "detector.1", relu=False)
    return osp.logits_to_heatmap(logits), F.normalize(desc, p=2, dim=1)


def kp(h, k):
    s = osp.kill_borders(osp.nms(h.float(), 3), 4)
    xy, _ = osp.select_keypoints(s[0], 0.0, k)
    return set(map(tuple, xy.tolist()))


VARIANTS = {
    "direct32": lambda name: direct,
    "F(2,3) all": lambda name: (lambda x, w, b: wino_conv(x, w, b, W2)),
    "F(4,3) all": lambda name: (lambda x, w, b: wino_conv(x, w, b, W4)),
    "F(4,3) conv1b only, F(2,3) rest": lambda name: (lambda x, w, b: wino_conv(x, w, b, W4 if name == "backbone.0.1" else W2)),
    "F(4,3) conv1b,2a,2b, F(2,3) rest": lambda name: (lambda x, w, b: wino_conv(
        x, w, b, W4 if name in ("backbone.0.1", "backbone.1.0", "backbone.1.1") else W2)),
    "F(4,3) blocks 0-2, F(2,3) rest": lambda name: (lambda x, w, b: wino_conv(
        x, w, b, W4 if name.startswith(("backbone.0", "backbone.1", "backbone.2")) else W2)),
}


def main():
    cases = [("vga seed 1234 img0", synthetic.synthetic_images(2, 480, 640, seed=1234)[:1], 1024),
             ("vga seed 1234 img1", synthetic.synthetic_images(2, 480, 640, seed=1234)[1:2], 1024),
             ("vga seed 41", synthetic.synthetic_images(1, 480, 640, seed=41), 1024),
             ("vga seed 77", synthetic.synthetic_images(1, 480, 640, seed=77), 1024),
             ("vga seed 5", synthetic.synthetic_images(1, 480, 640, seed=5), 1024),
             ("vga seed 6", synthetic.synthetic_images(1, 480, 640, seed=6), 1024),
             ("1024^2 seed 1234", synthetic.synthetic_images(1, 1024, 1024, seed=1234), 2048)]
    if len(sys.argv) > 1:
        cases = cases[: int(sys.argv[1])]
    worst = {n: [0.0, 0.0, 0] for n in VARIANTS}
    with torch.no_grad():
        for label, img, k in cases:
            h64, d64 = dense(img, lambda name: direct, torch.float64)
            ref_kp = None
            for name, sel in VARIANTS.items():
                h, d = dense(img, sel)
                herr = (h.double() - h64).abs().max().item()
                derr = (d.double() - d64).abs().max().item()
                kps = kp(h, k)
                if ref_kp is None:
                    ref_kp = kps
                flips = len(kps ^ ref_kp) // 2
                worst[name][0] = max(worst[name][0], herr)
                worst[name][1] = max(worst[name][1], derr)
                worst[name][2] += flips
                print(f"{label:22s} {name:34s} heat err {herr:.2e}  desc err {derr:.2e}  flips vs direct32 {flips}", flush=True)
    print("---- worst over cases (heat err, desc err, total flips)")
    for name, (a, b, c) in worst.items():
        print(f"{name:34s} {a:.2e} {b:.2e} {c}")


if __name__ == "__main__":
    main()

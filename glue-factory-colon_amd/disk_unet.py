"""The DISK network on MI355X: the U-Net behind `kornia.feature.DISK.heatmap_and_dense_descriptors`, which the reference
reaches at gluefactory/models/extractors/disk_kornia.py:24-47, on the HIP kernels of csrc/disk_unet.hip.

kornia's source and its pretrained weights are absent offline; the architecture is restated from kornia's published
code (kornia/feature/disk/disk.py, kornia/feature/disk/_unets/{unet,blocks}.py; see oracle/disk_unet.py for the
restatement the tests check against) -- NETWORK PARITY UNPINNED.  The module tree below exists to carry kornia's
parameter names (`unet.path_down.<i>.1.{1,3}.*`, `unet.path_up.<i>.conv.{1,3}.*`), so that a kornia DISK checkpoint
loads with `load_state_dict`; its torch modules are never called.

Data flow per call (NHWC fp32, b images of H x W, both divisible by 16):

    image NCHW -> [b,H,W,4]                                                     gfc_disk_nchw3_to_nhwc4
    down 0: conv(4 -> 16)            -> cat3[..., 64:80]   (f1, H)               gfc_disk_conv5x5
    down i: avg-pool, stats, conv    -> the slice of the concatenated tensor the matching up block reads
                                        (f2 -> cat2[..., 64:96] at H/2, f3 -> cat1[..., 64:128] at H/4,
                                         f4 -> cat0[..., 64:128] at H/8), f5 [b,H/16,W/16,64]
    up j:   bilinear x2 of the bottom path -> cat_j[..., 0:64]; stats over the whole cat_j; conv
    last up: 80 -> desc_dim + 1: descriptors [b,H,W,desc_dim] and the heat-map [b,H,W] are written to two arrays

No torch.cat, no NCHW <-> NHWC transposes between the layers; InstanceNorm + PReLU are applied while the convolution
stages its input (one statistics pass per layer is the only extra read of an activation).
"""
import torch
from torch import nn

from . import _native as nat

DOWN = (16, 32, 64, 64, 64)
UP = (64, 64, 64)


class _Conv(nn.Sequential):
    """kornia `Conv`: Sequential(norm, gate, dropout, Conv2d) -- parameter holders only."""

    def __init__(self, cin, cout, gated):
        super().__init__(nn.Identity(), nn.PReLU(cin) if gated else nn.Identity(), nn.Identity(),
                         nn.Conv2d(cin, cout, 5, padding=2))


class _Down(nn.Sequential):
    def __init__(self, cin, cout, first):
        super().__init__(nn.Identity(), _Conv(cin, cout, not first))


class _Up(nn.Module):
    def __init__(self, bottom, horizontal, cout):
        super().__init__()
        self.conv = _Conv(bottom + horizontal, cout, True)


class _Unet(nn.Module):
    def __init__(self, desc_dim):
        super().__init__()
        down = (3,) + DOWN
        self.path_down = nn.ModuleList([_Down(down[i], down[i + 1], i == 0) for i in range(5)])
        up = UP + (desc_dim + 1,)
        bot, hor = (DOWN[-1],) + up, down[-2::-1]
        self.path_up = nn.ModuleList([_Up(bot[i], hor[i], up[i]) for i in range(4)])


class _PackedLayer:
    def __init__(self, conv_holder, cin_run, device, st):
        lib = nat.lib()
        conv = conv_holder[3]
        gate = conv_holder[1]
        w = conv.weight.detach().to(device=device, dtype=torch.float32)
        cout, cin = int(w.shape[0]), int(w.shape[1])
        if cin_run != cin:  # the first layer runs on 4 input channels (RGB + a zero channel)
            w = torch.cat([w, w.new_zeros((cout, cin_run - cin, 5, 5))], 1)
        w = w.contiguous()
        self.cin, self.cout = cin_run, cout
        self.w = torch.empty((lib.gfc_disk_conv5x5_packed_floats(cout, cin_run),), device=device, dtype=torch.float32)
        nat.check(lib.gfc_disk_pack_conv5x5(nat.ptr(w), nat.ptr(self.w), cout, cin_run, st), "gfc_disk_pack_conv5x5")
        self.bias = conv.bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.slope = (gate.weight.detach().to(device=device, dtype=torch.float32).contiguous()
                      if isinstance(gate, nn.PReLU) else None)


class DiskUnet(nn.Module):
    """`heatmap_and_dense_descriptors(images)` of kornia's DISK on the native kernels; state-dict compatible with it."""

    def __init__(self, desc_dim: int = 128):
        super().__init__()
        self.desc_dim = int(desc_dim)
        self.unet = _Unet(self.desc_dim)
        self._packed = None
        self._ws = nat.Workspace()

    def load_state_dict(self, *args, **kwargs):
        ret = super().load_state_dict(*args, **kwargs)
        self._packed = None
        return ret

    def _load_from_state_dict(self, *args, **kwargs):
        # a PARENT's load_state_dict (DISK, TwoViewPipeline) recurses through this hook, not through the override
        # above: weights loaded that way must also drop the packed device copies of the previous ones
        self._packed = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._packed = None
        return super()._apply(fn, *args, **kwargs)

    def _pack(self, device):
        st = nat.stream_ptr(device)
        cin_run = (4,) + DOWN[:4]  # the first layer runs on RGB + a zero channel
        layers = [_PackedLayer(self.unet.path_down[i][1], cin_run[i], device, st) for i in range(5)]
        layers += [_PackedLayer(self.unet.path_up[i].conv, int(self.unet.path_up[i].conv[3].weight.shape[1]), device, st)
                   for i in range(4)]
        return {"device": device, "layers": layers}

    def ensure_packed(self, device):
        """Packed filters for `device`, built on the calling thread's current stream if they do not exist yet."""
        if self._packed is None or self._packed["device"] != device:
            self._packed = self._pack(device)
        return self._packed

    # ---- one Conv of the network: [statistics ->] convolution with normalisation + gate fused into its input stage ----
    def _conv(self, layer, x, b, h, w, y, ldy, gated, st):
        lib = nat.lib()
        mean = rstd = None
        if gated:
            stats = torch.empty((2, b, layer.cin), device=x.device, dtype=torch.float32)
            mean, rstd = stats[0], stats[1]
            ws = self._ws.get(lib.gfc_disk_instnorm_workspace_bytes(b, layer.cin), x.device)
            nat.check(lib.gfc_disk_instnorm_stats(nat.ptr(x), b, h, w, layer.cin, 1e-5, nat.ptr(mean), nat.ptr(rstd),
                                                  nat.ptr(ws), ws.numel(), st), "gfc_disk_instnorm_stats")
        nat.check(lib.gfc_disk_conv5x5(nat.ptr(x), nat.ptr(mean), nat.ptr(rstd), nat.ptr(layer.slope if gated else None),
                                       nat.ptr(layer.w), nat.ptr(layer.bias), y.data_ptr(), ldy, b, h, w, layer.cin,
                                       layer.cout, 0, layer.cout, st), "gfc_disk_conv5x5")

    def dense_nhwc(self, images):
        """images [b,3,H,W] on the GPU (H, W divisible by 16) -> heat-map [b,H,W], descriptors [b,H,W,desc_dim]."""
        nat.require_cuda(images, "images")
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"DISK expects RGB images [b,3,H,W], got {tuple(images.shape)}")
        b, _, H, W = images.shape
        if H % 16 or W % 16:
            raise ValueError(f"DISK: image size {(H, W)} is not divisible by 16 (pad_if_not_divisible)")
        if (H // 16) * (W // 16) < 2:  # torch's instance_norm refuses a 1 x 1 map too (no variance to normalise by)
            raise ValueError(f"DISK: image size {(H, W)} leaves one pixel at the coarsest level; InstanceNorm needs more than "
                             "1 spatial element")
        dev = images.device
        L = self.ensure_packed(dev)["layers"]
        lib, st = nat.lib(), nat.stream_ptr(dev)
        images = images.contiguous().float()
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
        d = self.desc_dim
        with torch.no_grad():
            x0 = new(b, H, W, 4)
            nat.check(lib.gfc_disk_nchw3_to_nhwc4(nat.ptr(images), b, H, W, nat.ptr(x0), st), "gfc_disk_nchw3_to_nhwc4")
            # concatenated tensors of the up path, [bottom_big (64) | horizontal]: cat[j] is read by up block j
            hor_c = (64, 64, 32, 16)
            cat = [new(b, H >> (3 - j), W >> (3 - j), 64 + hor_c[j]) for j in range(4)]
            # ---- down path: f_{i+1} goes into the slice of the tensor its up block reads ----
            self._conv(L[0], x0, b, H, W, cat[3][..., 64:], cat[3].shape[-1], False, st)  # f1
            src, src_ld, c = cat[3][..., 64:], cat[3].shape[-1], 16
            f5 = None
            for i in range(1, 5):
                h, w = H >> i, W >> i
                pooled = new(b, h, w, c)
                nat.check(lib.gfc_disk_avgpool2(src.data_ptr(), src_ld, b, h * 2, w * 2, c, nat.ptr(pooled), st),
                          "gfc_disk_avgpool2")
                if i < 4:
                    dst = cat[3 - i]
                    self._conv(L[i], pooled, b, h, w, dst[..., 64:], dst.shape[-1], True, st)
                    src, src_ld, c = dst[..., 64:], dst.shape[-1], L[i].cout
                else:
                    f5 = new(b, h, w, 64)
                    self._conv(L[4], pooled, b, h, w, f5, 64, True, st)
            # ---- up path ----
            bot, bh, bw = f5, H >> 4, W >> 4
            heat = desc = None
            for j in range(4):
                nat.check(lib.gfc_disk_upsample2(nat.ptr(bot), b, bh, bw, 64, nat.ptr(cat[j]), cat[j].shape[-1], st),
                          "gfc_disk_upsample2")
                bh, bw = bh * 2, bw * 2
                if j < 3:
                    out = new(b, bh, bw, 64)
                    self._conv(L[5 + j], cat[j], b, bh, bw, out, 64, True, st)
                    bot = out
                else:
                    # desc_dim + 1 output channels: descriptors and heat-map go to two arrays (two launches over the same
                    # statistics: the heat-map channel is a 32-wide block of its own either way)
                    desc, heat = new(b, H, W, d), new(b, H, W)
                    self._conv_split(L[8], cat[3], b, H, W, desc, heat, st)
        return heat, desc

    def _conv_split(self, layer, x, b, h, w, desc, heat, st):
        lib = nat.lib()
        d = self.desc_dim
        stats = torch.empty((2, b, layer.cin), device=x.device, dtype=torch.float32)
        ws = self._ws.get(lib.gfc_disk_instnorm_workspace_bytes(b, layer.cin), x.device)
        nat.check(lib.gfc_disk_instnorm_stats(nat.ptr(x), b, h, w, layer.cin, 1e-5, nat.ptr(stats[0]), nat.ptr(stats[1]),
                                              nat.ptr(ws), ws.numel(), st), "gfc_disk_instnorm_stats")
        for y, ldy, first, count in ((desc, d, 0, d), (heat, 1, d, 1)):  # descriptors [0,d), heat-map channel d
            nat.check(lib.gfc_disk_conv5x5(nat.ptr(x), nat.ptr(stats[0]), nat.ptr(stats[1]), nat.ptr(layer.slope),
                                           nat.ptr(layer.w), nat.ptr(layer.bias), nat.ptr(y), ldy, b, h, w, layer.cin,
                                           layer.cout, first, count, st), "gfc_disk_conv5x5")

    def heatmap_and_dense_descriptors(self, images):
        """kornia's contract: (heat-maps [b,1,H,W], descriptors [b,desc_dim,H,W]) -- views of the NHWC arrays."""
        heat, desc = self.dense_nhwc(images)
        return heat.unsqueeze(1), desc.permute(0, 3, 1, 2)

    forward = heatmap_and_dense_descriptors

"""CPU emulation of Winograd F(2x2,3x3) in fp32 through the whole SuperPoint-open stack (name-seeded weights, one VGA
image): heat-map / descriptor error against a float64 evaluation next to the direct fp32 convolution's, and the
number of key-point flips among the top 1024.  Decision experiment for csrc/conv_wino.hip:
    direct32 heat err 3.2e-06 desc err 1.6e-06 | wino32 heat err 2.4e-06 desc err 1.1e-06 | flips 0 / 0 / 0
"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from glue_factory_colon_amd import weights, synthetic
from oracle import superpoint as osp
torch.set_num_threads(8)
sd = weights.superpoint_open_state_dict(0)
img = synthetic.synthetic_images(2, 480, 640, seed=1234)[:1]

BT = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float64)
G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float64)
AT = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float64)

def wino_conv(x, w, b):
    """x [B,C,H,W] fp32, 3x3 pad 1, F(2x2,3x3) emulated in fp32"""
    Bn, C, H, W = x.shape
    Co = w.shape[0]
    U = (G @ w.double() @ G.T).float()              # [Co,C,4,4]  (pack time, in double, rounded once)
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    # tiles 4x4 stride 2
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)          # [B,C,th,tw,4,4]
    bt = BT.float()
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", bt, t, bt)   # B^T d B  (fp32 adds)
    M = torch.einsum("ocil,bcyxil->boyxil", U, V)   # fp32
    at = AT.float()
    Y = torch.einsum("ij,boyxjk,lk->boyxil", at, M, at)   # [B,Co,th,tw,2,2]
    th, tw = Y.shape[2], Y.shape[3]
    Y = Y.permute(0,1,2,4,3,5).reshape(Bn, Co, th*2, tw*2)[:, :, :H, :W]
    return Y + b.view(1,-1,1,1)

def dense(sd, image, conv, dt=torch.float32):
    x = image.to(dt)
    def block(x, prefix, relu=True, c=conv):
        w = sd[prefix + ".conv.weight"].to(dt)
        if w.shape[-1] == 3:
            x = c(x, w, sd[prefix + ".conv.bias"].to(dt))
        else:
            x = F.conv2d(x, w, sd[prefix + ".conv.bias"].to(dt))
        if relu: x = F.relu(x)
        return F.batch_norm(x, sd[prefix + ".bn.running_mean"].to(dt), sd[prefix + ".bn.running_var"].to(dt),
                            sd[prefix + ".bn.weight"].to(dt), sd[prefix + ".bn.bias"].to(dt), training=False, eps=0.001)
    direct = lambda x, w, b: F.conv2d(x, w, b, padding=1)
    for bk in range(4):
        x = block(x, f"backbone.{bk}.0", c=direct if bk == 0 else conv)   # conv1a (cin=1) stays direct
        x = block(x, f"backbone.{bk}.1")
        if bk < 3: x = F.max_pool2d(x, 2, 2)
    desc = block(block(x, "descriptor.0"), "descriptor.1", relu=False)
    logits = block(block(x, "detector.0"), "detector.1", relu=False)
    return osp.logits_to_heatmap(logits), F.normalize(desc, p=2, dim=1), x

direct = lambda x, w, b: F.conv2d(x, w, b, padding=1)
with torch.no_grad():
    h64, d64, x64 = dense(sd, img, direct, torch.float64)
    h32, d32, x32 = dense(sd, img, direct)
    hw, dw, xw = dense(sd, img, wino_conv)
def kp(h):
    s = osp.kill_borders(osp.nms(h.float(), 3), 4)
    xy, v = osp.select_keypoints(s[0], 0.0, 1024)
    return set(map(tuple, xy.tolist()))
for name, (h, d, x) in {"direct32": (h32, d32, x32), "wino32": (hw, dw, xw)}.items():
    print(name, "heat err", (h.double() - h64).abs().max().item(), "desc err", (d.double() - d64).abs().max().item(),
          "feat err", (x.double() - x64).abs().max().item(), "feat max", x64.abs().max().item())
k64, k32, kw = kp(h64), kp(h32), kp(hw)
print("kp flips direct32 vs 64:", len(k64 ^ k32), " wino vs 64:", len(k64 ^ kw), " wino vs direct32:", len(k32 ^ kw))

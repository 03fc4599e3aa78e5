import sys, time, os, torch
sys.path.insert(0, os.getcwd())
from glue_factory_colon_amd import weights
from oracle import lightglue as olg
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), "affinity", len(os.sched_getaffinity(0)))
sd = weights.lightglue_state_dict(0)
g = torch.Generator().manual_seed(0)
B, K = 4, 1024
kp = torch.rand((B, K, 2), generator=g) * 480
d = torch.nn.functional.normalize(torch.randn((B, K, 256), generator=g), dim=-1)
size = torch.tensor([[640.0, 480.0]]).expand(B, 2)
for n in (None, 64, 32, 16, 8):
    if n: torch.set_num_threads(n)
    t = time.time(); olg.match(sd, kp, kp.flip(1), d, d.flip(1), size, size, filter_threshold=0.1); dt = time.time() - t
    print("threads", torch.get_num_threads(), f"{dt / B:.2f} s/pair", flush=True)

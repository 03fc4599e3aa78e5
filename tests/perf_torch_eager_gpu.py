"""Not a test (no test_ prefix): times the oracle -- the PyTorch restatement of the reference path -- with its
tensors on the MI355X (PyTorch-ROCm eager: MIOpen convolutions, rocBLAS/hipBLASLt linears, SDPA), i.e. what the
reference's own modules do on this GPU, next to this package's path.  Information for DESIGN.md section 5."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import synthetic, weights  # noqa: E402
from oracle import lightglue as olg  # noqa: E402
from oracle import superpoint as osp  # noqa: E402


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda", 0)
    h, w, k = 480, 640, 1024
    v0, v1 = synthetic.synthetic_pairs(b, h, w, seed=1234, device=dev)
    sd_sp = {kk: v.to(dev) for kk, v in weights.superpoint_open_state_dict(0).items()}
    sd_lg = {kk: v.to(dev) for kk, v in weights.lightglue_state_dict(0).items()}
    size = torch.tensor([[float(w), float(h)]] * b, device=dev)

    def run():
        a = osp.extract(sd_sp, v0, "open", nms_radius=3, max_num_keypoints=k, detection_threshold=0.0)
        c = osp.extract(sd_sp, v1, "open", nms_radius=3, max_num_keypoints=k, detection_threshold=0.0)
        out = olg.match(sd_lg, torch.stack(a["keypoints"]), torch.stack(c["keypoints"]),
                        torch.stack(a["descriptors"]), torch.stack(c["descriptors"]), size, size, filter_threshold=0.1)
        return out

    with torch.no_grad():
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({"torch_eager_pairs_per_s": round(b * iters / dt, 2), "pairs_per_call": b, "iters": iters,
                      "first_call_s": round(first, 1), "matches": int((out["matches0"] >= 0).sum())}), flush=True)


if __name__ == "__main__":
    main()

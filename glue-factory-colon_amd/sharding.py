"""Data-parallel sharding of image pairs over the GPUs of one node.

The reference evaluates on a single device (gluefactory/utils/export_predictions.py:34-35);
pairs are independent (gluefactory/models/two_view_pipeline.py:278-339), so the MI355X build
runs one process per GPU, gives each rank a contiguous block (synthetic, equal sizes) or a
round-robin slice (HPatches-style variable sizes) of the pair list, replicates the weights
(52.6 MB) and needs exactly ONE collective: a gather of fixed-size per-pair records to rank 0
at the end (RCCL over xGMI: one direct peer write per rank on the fully connected mesh; no ring).
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns
    (rank, world_size, local_rank).  backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def contiguous_shard(n_items: int, rank: int, world: int) -> range:
    """Block partition; the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def round_robin_shard(n_items: int, rank: int, world: int, group: int = 1):
    """Items dealt out round-robin, `group` consecutive items at a time (group 5 keeps the five pairs of an HPatches
    sequence -- which share their view-0 image -- on one rank, next to each other)."""
    if group <= 1:
        return range(rank, n_items, world)
    return [i for i in range(n_items) if (i // group) % world == rank]


def gather_records(records: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather one equally-shaped record tensor per rank to `dst` (single collective).
    Returns the list on dst, None elsewhere; a no-op list without a process group (with a one-rank group the collective
    still runs: tests/test_gpu_rccl.py exercises the RCCL calls of the path on a one-GPU box that way)."""
    if not dist.is_initialized():
        return [records]
    world, rank = dist.get_world_size(), dist.get_rank()
    records = records.contiguous()
    out = [torch.empty_like(records) for _ in range(world)] if rank == dst else None
    dist.gather(records, out, dst=dst)
    return out


def pack_pair_records(pred: dict, max_kpts: int) -> torch.Tensor:
    """Fixed-size per-pair record for the final gather: [B, 2 + 6*K] float32 =
    (n_matches, n_keypoints0, keypoints0 xy, keypoints1 xy, matches0, matching_scores0), padded to K."""
    b = pred["matches0"].shape[0]
    k = max_kpts
    rec = torch.zeros((b, 2 + 6 * k), device=pred["matches0"].device, dtype=torch.float32)
    m0 = pred["matches0"]
    n0 = min(m0.shape[1], k)
    n1 = min(pred["keypoints1"].shape[1], k)
    rec[:, 0] = (m0 >= 0).sum(1).float()
    rec[:, 1] = float(m0.shape[1])
    rec[:, 2:2 + 2 * n0] = pred["keypoints0"][:, :n0].reshape(b, -1)
    rec[:, 2 + 2 * k:2 + 2 * k + 2 * n1] = pred["keypoints1"][:, :n1].reshape(b, -1)
    rec[:, 2 + 4 * k:2 + 4 * k + n0] = m0[:, :n0].float()
    rec[:, 2 + 5 * k:2 + 5 * k + n0] = pred["matching_scores0"][:, :n0]
    return rec

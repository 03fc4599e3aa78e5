"""The RCCL calls of the path on ONE GPU (one gather of per-pair records to rank 0; export_predictions' sharded
mode: an int64 all-gather of block sizes, a uint8 gather of the record blocks, an int32 MAX all-reduce as failure flag;
bench.py: a float64 MAX all-reduce of the elapsed time, barriers).  A multi-GPU node is not available to the builder, and
RCCL refuses two ranks on one device, so the N > 1 logic is covered on gloo (tests/test_host_cpu.py, world 2 and 8) and THIS
test runs the same collectives, with the same dtypes and devices, through backend "nccl" (= RCCL) in a one-rank group: an
unsupported dtype / device combination or a missing RCCL entry point fails here instead of in the first 8-GPU run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_one_rank_group_runs_every_collective_of_the_path(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, socket, numpy as np, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from glue_factory_colon_amd import sharding\n"
        "from glue_factory_colon_amd import export_predictions as ep\n"
        "with socket.socket() as so:\n"
        "    so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')\n"
        "os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group(backend='nccl', rank=0, world_size=1)\n"
        "dev = torch.device('cuda', 0)\n"
        "k = 16\n"
        "pred = {'matches0': torch.randint(-1, k, (4, k), device=dev), 'keypoints0': torch.rand(4, k, 2, device=dev),\n"
        "        'keypoints1': torch.rand(4, k, 2, device=dev), 'matching_scores0': torch.rand(4, k, device=dev)}\n"
        "rec = sharding.pack_pair_records(pred, k)\n"
        "out = sharding.gather_records(rec)                       # dist.gather of float32 records (bench.py, SURVEY 8e)\n"
        "assert len(out) == 1 and torch.equal(out[0], rec)\n"
        "t = torch.tensor([1.25], device=dev, dtype=torch.float64)\n"
        "dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t) == 1.25   # bench.py: max over ranks of the elapsed time\n"
        "dist.barrier()\n"
        "local = [(0, 'a/2.ppm', {'matches0': np.arange(5, dtype=np.int64), 'keypoints0': np.ones((5, 2), np.float32)}),\n"
        "         (1, 'a/3.ppm', {'matches0': np.arange(3, dtype=np.int64), 'keypoints0': np.zeros((3, 2), np.float32)})]\n"
        "blocks, failed = ep._gather_to_rank0(local, False, 0, 1, 'cuda')  # int64 all_gather + uint8 gather on the device\n"
        "assert not failed and len(blocks) == 1\n"
        "ent = ep._decode_blocks(blocks)\n"
        "assert [(i, n) for i, n, _ in ent] == [(0, 'a/2.ppm'), (1, 'a/3.ppm')]\n"
        "assert np.array_equal(ent[0][2]['matches0'], np.arange(5)) and ent[1][2]['keypoints0'].shape == (3, 2)\n"
        "assert ep._any_rank_failed(False, 'cuda') is False and ep._any_rank_failed(True, 'cuda') is True  # int32 MAX all-reduce\n"
        "blocks, failed = ep._gather_to_rank0([], True, 0, 1, 'cuda')      # a failing rank: the size all-gather carries the flag\n"
        "assert failed and blocks is None\n"
        "dist.destroy_process_group()\n"
        "print('RCCL_ONE_RANK_OK')\n")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]

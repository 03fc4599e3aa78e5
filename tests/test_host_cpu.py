"""CPU-side checks (no GPU, no compute calls into the HIP library): the C-ABI library loads and
exports every declared symbol, the boundary modules mirror the reference's module contract, the
product path refuses CPU tensors (no fallback), weight packing order, sharding helpers."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from glue_factory_colon_amd import _native as nat
from glue_factory_colon_amd import base_model, lightglue, lightglue_pretrained, registry, sharding, superpoint
from glue_factory_colon_amd import superpoint_open, synthetic, weights
from glue_factory_colon_amd._superpoint_common import fold_bn
from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gfc_amd.h")).read()
    declared = set(re.findall(r"\b(gfc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 18
    lib = nat.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gfc_amd.h but not exported"
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)
    assert b"gfx950" in lib.gfc_version()


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "glue-factory-colon_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


def test_conf_merge_and_contract():
    m = superpoint_open.SuperPoint({"weights": "synthetic", "max_num_keypoints": 77, "some_unknown_key": 1})
    assert m.conf.max_num_keypoints == 77 and m.conf.nms_radius == 4 and m.conf.remove_borders == 4
    assert m.conf.trainable is True and m.conf.name is None  # BaseModel defaults (base_model.py:54-59)
    assert m.is_initialized()
    with pytest.raises(AssertionError, match="Missing key image"):
        m({"not_image": torch.zeros(1)})
    with pytest.raises(nat.NativeError, match="no CPU implementation"):
        m({"image": torch.zeros(1, 1, 32, 32)})
    with pytest.raises(NotImplementedError):
        m.loss({}, {})
    un = superpoint_open.SuperPoint({})
    assert not un.is_initialized()
    un.load_state_dict(weights.superpoint_open_state_dict(1))
    assert un.is_initialized()
    with pytest.raises(FileNotFoundError):
        superpoint_open.SuperPoint({"weights": "/nonexistent/superpoint_v6_from_tf.pth"})


def test_state_dict_key_layouts_match_reference():
    sd = weights.superpoint_open_state_dict(0)
    assert len(sd) == 84 and sum(v.numel() for k, v in sd.items() if "num_batches" not in k) > 1_300_000
    superpoint_open.SuperPoint({}).load_state_dict(sd, strict=True)
    sd = weights.superpoint_state_dict(0)
    assert len(sd) == 24
    superpoint.SuperPoint({}).load_state_dict(sd, strict=True)
    sd = weights.lightglue_state_dict(0)
    assert len(sd) == 251  # + confidence_thresholds buffer = 252
    m = lightglue.LightGlue({})
    missing = m.load_state_dict(sd, strict=False)
    assert missing.missing_keys == ["confidence_thresholds"] and not missing.unexpected_keys
    assert sum(v.numel() for v in m.state_dict().values()) == 11_851_610
    # legacy checkpoint key names (lightglue.py:394-401)
    legacy = {k.replace("transformers.3.self_attn", "self_attn.3").replace("transformers.3.cross_attn", "cross_attn.3"): v
              for k, v in sd.items()}
    assert any(k.startswith("self_attn.3") for k in legacy)
    missing = lightglue.LightGlue({}).load_state_dict(legacy, strict=False)
    assert missing.missing_keys == ["confidence_thresholds"] and not missing.unexpected_keys
    assert torch.allclose(m.confidence_thresholds[0], torch.tensor(0.9))


def test_weights_are_deterministic():
    a, b = weights.lightglue_state_dict(0), weights.lightglue_state_dict(0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = weights.lightglue_state_dict(1)
    assert not torch.equal(a["posenc.Wr.weight"], c["posenc.Wr.weight"])
    v0, v1 = synthetic.synthetic_pairs(1, 64, 96, seed=5, dx=16, dy=8)
    assert torch.equal(v1[0, 0, 8:, 16:], v0[0, 0, :-8, :-16])


def test_registry_and_pipeline_contract():
    assert registry.get_model("extractors.superpoint_open") is superpoint_open.SuperPoint
    assert registry.get_model("gluefactory_nonfree.superpoint") is superpoint.SuperPoint
    assert registry.get_model("matchers.lightglue") is lightglue.LightGlue
    assert registry.get_model("matchers.lightglue_pretrained") is lightglue_pretrained.LightGlue
    assert registry.get_model("glue_factory_colon_amd.lightglue") is lightglue.LightGlue
    with pytest.raises(RuntimeError, match="not found"):
        registry.get_model("extractors.aliked")
    conf = {"extractor": {"name": "gluefactory_nonfree.superpoint", "weights": "synthetic", "max_num_keypoints": 1024,
                          "detection_threshold": 0.0, "nms_radius": 3},
            "matcher": {"name": "matchers.lightglue_pretrained", "features": "superpoint", "weights": "synthetic",
                        "depth_confidence": -1, "width_confidence": -1, "filter_threshold": 0.1}}
    pipe = TwoViewPipeline(conf)  # the superpoint+lightglue-official.yaml model block
    assert pipe.is_initialized() and pipe.extractor.conf.legacy_sampling is True
    with pytest.raises(AssertionError, match="Missing key view1"):
        pipe({"view0": {}})
    with pytest.raises(NotImplementedError):
        TwoViewPipeline({"solver": {"name": "homography_est"}})
    un = TwoViewPipeline({"extractor": {"name": "extractors.superpoint_open"}, "matcher": {"name": "matchers.lightglue"}})
    assert not un.is_initialized()


def test_lightglue_rejects_out_of_scope_configs():
    assert lightglue.LightGlue({"add_scale_ori": True}).posenc.Wr.weight.shape == (32, 4)  # built (SIFT-style inputs)
    with pytest.raises(NotImplementedError):
        lightglue.LightGlue({"descriptor_dim": 128})
    m = lightglue.LightGlue({"weights": "synthetic", "depth_confidence": 0.95}).eval()
    d = {"keypoints0": torch.zeros(1, 4, 2), "keypoints1": torch.zeros(1, 4, 2), "descriptors0": torch.zeros(1, 4, 256),
         "descriptors1": torch.zeros(1, 4, 256)}
    with pytest.raises(nat.NativeError, match="no CPU implementation"):  # adaptive path is built, but GPU only
        m({**d, "view0": {"image_size": torch.ones(1, 2)}, "view1": {"image_size": torch.ones(1, 2)}})
    with pytest.raises(NotImplementedError, match="training"):
        lightglue.LightGlue({"weights": "synthetic"}).train()(d)
    with pytest.raises(AssertionError, match="Missing key descriptors1"):
        m({k: v for k, v in d.items() if k != "descriptors1"})


def test_wqkv_row_permutation_matches_unflatten():
    """packed row s*256 + h*64 + d must hold state-dict row h*192 + d*3 + s (lightglue.py:157-159)."""
    d, h, dh = 256, 4, 64
    w = torch.arange(3 * d, dtype=torch.float32)[:, None].expand(-1, 2)
    qkv = w[:, 0].view(h, dh, 3)  # unflatten(-1, (h, -1, 3)) of the output channel axis
    idx = torch.arange(3 * d)
    s_, rem = idx // d, idx % d
    src = (rem // dh) * (3 * dh) + (rem % dh) * 3 + s_
    packed = w[src, 0].view(3, h, dh)
    for s in range(3):
        assert torch.equal(packed[s], qkv[..., s])


def test_fold_bn_matches_torch_eval_batchnorm():
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 8, 5, 5), generator=g)
    w, b = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g)
    mean, var = torch.randn(8, generator=g), torch.rand(8, generator=g) + 0.2
    a, c = fold_bn(w, b, mean, var, 1e-3)
    ref = torch.nn.functional.batch_norm(x, mean, var, w, b, False, 0.0, 1e-3)
    assert (x * a[None, :, None, None] + c[None, :, None, None] - ref).abs().max() < 1e-6


def test_shard_partitions():
    for n, w in ((256, 8), (540, 8), (7, 3), (2, 4)):
        blocks = [list(sharding.contiguous_shard(n, r, w)) for r in range(w)]
        assert sorted(sum(blocks, [])) == list(range(n))
        assert max(map(len, blocks)) - min(map(len, blocks)) <= 1
        rr = [list(sharding.round_robin_shard(n, r, w)) for r in range(w)]
        assert sorted(sum(rr, [])) == list(range(n))
        g5 = [list(sharding.round_robin_shard(n, r, w, 5)) for r in range(w)]  # whole groups of five per rank
        assert sorted(sum(g5, [])) == list(range(n))
        assert all(len({i // 5 for i in chunk}) * 5 >= len(chunk) for chunk in g5)
        assert all(i // 5 % w == r for r, chunk in enumerate(g5) for i in chunk)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])


def test_gather_records_world2_gloo(tmp_path):
    """The only collective of the path: one gather of fixed-size records, 2 ranks on gloo."""
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from glue_factory_colon_amd import sharding\n"
        "rank, world, _ = sharding.init_from_env('gloo')\n"
        "pairs = list(sharding.contiguous_shard(6, rank, world))\n"
        "pred = {'matches0': torch.tensor([[p, -1, 2] for p in pairs]), 'keypoints0': torch.ones(len(pairs), 3, 2) * rank,\n"
        "        'keypoints1': torch.ones(len(pairs), 3, 2), 'matching_scores0': torch.full((len(pairs), 3), 0.5)}\n"
        "rec = sharding.pack_pair_records(pred, 4)\n"
        "out = sharding.gather_records(rec)\n"
        "if rank == 0:\n"
        "    allrec = torch.cat(out)\n"
        "    assert allrec.shape == (6, 26), allrec.shape\n"
        "    assert allrec[:, 2 + 16].tolist() == [0, 1, 2, 3, 4, 5]\n"
        "    assert allrec[:, 0].tolist() == [2.0] * 6\n"
        "    print('GATHER_OK')\n"
        "else:\n"
        "    assert out is None\n")
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0 and "GATHER_OK" in r.stdout, r.stdout + r.stderr


def test_base_model_alias():
    assert issubclass(superpoint_open.SuperPoint, base_model.BaseModel)
    assert superpoint_open.__main_model__ is superpoint_open.SuperPoint
    assert lightglue.__main_model__ is lightglue.LightGlue and not issubclass(lightglue.LightGlue, base_model.BaseModel)


@pytest.mark.skipif(not os.path.isdir("/root/reference/gluefactory"), reason="reference checkout not present")
def test_dropin_inside_reference_registry():
    """Build container only: the REFERENCE's get_model / TwoViewPipeline resolve and construct this package's
    modules from a config block (subclassing the reference's own BaseModel), and report initialised."""
    code = (
        "import sys\n"
        f"sys.path[:0] = [{os.path.join(ROOT, 'tests', 'golden', '_standins')!r}, '/root/reference', {ROOT!r}]\n"
        "sys.dont_write_bytecode = True\n"
        "from gluefactory.models import get_model\n"
        "from gluefactory.models.base_model import BaseModel\n"
        "from gluefactory.models.two_view_pipeline import TwoViewPipeline\n"
        "import glue_factory_colon_amd.base_model as bm\n"
        "assert bm.USING_REFERENCE_BASE and bm.BaseModel is BaseModel\n"
        "ext = get_model('glue_factory_colon_amd.superpoint_open')\n"
        "mat = get_model('glue_factory_colon_amd.lightglue')\n"
        "assert issubclass(ext, BaseModel) and mat.__name__ == 'LightGlue'\n"
        "assert get_model('glue_factory_colon_amd.superpoint').__module__ == 'glue_factory_colon_amd.superpoint'\n"
        "assert issubclass(get_model('glue_factory_colon_amd.lightglue_pretrained'), BaseModel)\n"
        "pipe = TwoViewPipeline({'extractor': {'name': 'glue_factory_colon_amd.superpoint_open', 'weights': 'synthetic',\n"
        "    'max_num_keypoints': 1024, 'detection_threshold': 0.0, 'nms_radius': 3},\n"
        "    'matcher': {'name': 'glue_factory_colon_amd.lightglue_pretrained', 'features': 'superpoint',\n"
        "    'weights': 'synthetic', 'filter_threshold': 0.1}}).eval()\n"
        "assert pipe.extractor.conf.max_num_keypoints == 1024 and pipe.extractor.conf.remove_borders == 4\n"
        "assert pipe.is_initialized()\n"
        "# this package's pipeline class (the one with forward_pairs, for export_predictions(pair_batch=N)) by name\n"
        "P = get_model('glue_factory_colon_amd.two_view_pipeline')\n"
        "assert issubclass(P, BaseModel) and hasattr(P, 'forward_pairs') and not hasattr(TwoViewPipeline, 'forward_pairs')\n"
        "p2 = P({'extractor': {'name': 'glue_factory_colon_amd.superpoint', 'weights': 'synthetic'},\n"
        "        'matcher': {'name': 'glue_factory_colon_amd.lightglue', 'weights': 'synthetic'}}).eval()\n"
        "assert p2.is_initialized() and hasattr(p2.extractor, 'forward_views') and hasattr(p2.matcher, 'forward_pairs')\n"
        "print('DROPIN_OK')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stdout + r.stderr


def test_bench_multiprocess_plumbing_rehearsal():
    """bench.py --gpus 2 launched exactly as the driver does (torch.distributed.run), with --rehearse-cpu:
    rendezvous on 127.0.0.1, barriers, MAX all-reduce of the elapsed time, the final gather, one JSON line."""
    import json

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--pairs", "4", "--rehearse-cpu", "--cpu-pairs", "1", "--cpu-iters", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout + r.stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["pairs_gathered"] == 8 and out["matches_per_rank"] == [10, 11]
    # the N > 1 line carries the CPU baseline too (rank 0 times it after the gather; bounded sample here)
    base = out["cpu_baseline"]
    assert base is not None and base["value"] > 0 and base["kind"] == "port" and base["cores"] >= 1


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as a plain command (the form the driver uses): the parent starts the two rank
    processes itself (free port on 127.0.0.1, RANK / LOCAL_RANK / WORLD_SIZE set), relays exactly one JSON line on
    stdout and exits with the ranks' status.  A failing rank makes the parent fail."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs",
           "4", "--rehearse-cpu", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and len(lines) == 1 and lines[0].startswith("{"), r.stdout + r.stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["pairs_gathered"] == 8 and out["matches_per_rank"] == [10, 11]
    # mismatch between --gpus and an inherited WORLD_SIZE is an error, not a silent single-rank run
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0


def test_export_predictions_sharded_world2_gloo(tmp_path):
    """HPatches-style export shared out over 2 ranks (round-robin over the pair list, ONE gather of the ranks' record
    blocks to rank 0, which writes the file -- SURVEY.md 8e; no part files): the file equals the single-process export -- same record names in loader order,
    same arrays, key points un-scaled by 1 / scales (reference utils/export_predictions.py:36-85).  Fake model on CPU."""
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, torch, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from glue_factory_colon_amd import sharding\n"
        "from glue_factory_colon_amd.export_predictions import export_predictions, load_predictions\n"
        "class Fake(torch.nn.Module):\n"
        "    def forward(self, data):\n"
        "        k0 = data['view0']['x'] * 2 + 1\n"
        "        return {'keypoints0': k0, 'keypoints1': data['view1']['x'], 'matches0': (k0[..., 0] > 3).long() - 1,\n"
        "                'matching_scores0': k0[..., 1], 'unused': k0}\n"
        "def item(i):\n"
        "    g = torch.Generator().manual_seed(i)\n"
        "    n = 5 + i % 3\n"
        "    v = lambda: {'x': torch.rand((1, n, 2), generator=g) * 9, 'scales': torch.tensor([[0.5 + 0.1 * i, 0.75]])}\n"
        "    return {'name': [f'v_seq{i // 3}/{i % 3 + 2}.ppm'], 'view0': v(), 'view1': v()}\n"
        "items = [item(i) for i in range(7)]\n"
        "items.append({**item(1)})  # duplicate name: the first occurrence wins\n"
        "keys = ['keypoints0', 'keypoints1', 'matches0', 'matching_scores0']\n"
        "rank, world, _ = sharding.init_from_env('gloo')\n"
        "out = sys.argv[1]\n"
        "export_predictions(items, Fake(), out + '/sharded.npz', keys=keys)\n"
        "export_predictions(items, Fake(), out + '/sharded_pb.npz', keys=keys, pair_batch=2)  # chunks of two pairs per rank\n"
        "if rank == 0:\n"
        "    c = load_predictions(out + '/sharded_pb.npz'); a0 = load_predictions(out + '/sharded.npz')\n"
        "    assert list(c) == list(a0) and all(np.array_equal(c[n][k], a0[n][k]) for n in c for k in c[n])\n"
        "    export_predictions(items, Fake(), out + '/single.npz', keys=keys, rank=0, world=1)\n"
        "    a, b = load_predictions(out + '/sharded.npz'), load_predictions(out + '/single.npz')\n"
        "    assert list(a) == list(b) == [f'v_seq{i // 3}/{i % 3 + 2}.ppm' for i in range(7)], list(a)\n"
        "    for n in a:\n"
        "        assert sorted(a[n]) == sorted(keys)\n"
        "        for k in keys:\n"
        "            assert np.array_equal(a[n][k], b[n][k]), (n, k)\n"
        "    it = items[4]\n"
        "    ref = (it['view0']['x'][0] * 2 + 1) * (1.0 / it['view0']['scales'])\n"
        "    assert np.array_equal(a[it['name'][0]]['keypoints0'], ref.numpy())\n"
        "    import os; assert not [f for f in os.listdir(out) if '.part' in f]\n"
        "    print('SHARDED_EXPORT_OK')\n"
        "torch.distributed.barrier()\n")
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0 and "SHARDED_EXPORT_OK" in r.stdout, r.stdout + r.stderr


def test_export_predictions_sharded_failure_does_not_hang(tmp_path):
    """One rank's model raises: every rank leaves export_predictions with an exception (the failing one with its own, the
    other with "another rank failed") instead of waiting in the gather, and no prediction file is written."""
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from glue_factory_colon_amd import sharding\n"
        "from glue_factory_colon_amd.export_predictions import export_predictions\n"
        "rank, world, _ = sharding.init_from_env('gloo')\n"
        "class Fake(torch.nn.Module):\n"
        "    def forward(self, data):\n"
        "        if rank == 1:\n"
        "            raise ValueError('boom on rank 1')\n"
        "        return {'keypoints0': data['view0']['x']}\n"
        "items = [{'name': [f's/{i}.ppm'], 'view0': {'x': torch.ones(1, 3, 2), 'scales': torch.ones(1, 2)}} for i in range(4)]\n"
        "try:\n"
        "    export_predictions(items, Fake(), sys.argv[1] + '/p.npz')\n"
        "    msg = f'NO_EXCEPTION {rank}'\n"
        "except ValueError as e:\n"
        "    msg = f'OWN_FAILURE {rank} {e}'\n"
        "except RuntimeError as e:\n"
        "    msg = f'TOLD {rank} {e}'\n"
        "open(sys.argv[1] + f'/rank{rank}.txt', 'w').write(msg)  # (the ranks' stdout lines may interleave)\n"
        "assert not os.path.exists(sys.argv[1] + '/p.npz')\n"
        "torch.distributed.barrier()\n")
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    said = [(tmp_path / f"rank{i}.txt").read_text() for i in range(2)]
    assert said[1] == "OWN_FAILURE 1 boom on rank 1", said
    assert said[0].startswith("TOLD 0") and "another rank failed" in said[0], said


def test_hdf5_prediction_file_without_h5py(tmp_path):
    """`predictions.h5` in the reference's layout (utils/export_predictions.py:81-90: group per "<seq>/<idx>.ppm",
    dataset per key) written and read through the HDF5 C library bound by ctypes (h5py is not installed here);
    export -> CacheLoader round trip on that container."""
    from glue_factory_colon_amd import _hdf5, cache_loader
    from glue_factory_colon_amd.export_predictions import export_predictions, load_predictions

    if not _hdf5.available():
        pytest.skip("no HDF5 C library in this environment")
    g = torch.Generator().manual_seed(0)
    recs = {"v_a/2.ppm": {"keypoints0": torch.rand((7, 2), generator=g).numpy(), "matches0": np.arange(7, dtype=np.int64) - 1,
                          "scores_half": torch.rand((7,), generator=g).numpy().astype(np.float16),
                          "flag": np.array([True, False, True]), "empty": np.zeros((0, 2), np.float32),
                          "scalar": np.float32(2.5)},
            "v_a/3.ppm": {"keypoints0": np.zeros((0, 2), np.float32)},
            "i_b/6.ppm": {"d": np.arange(12, dtype=np.float64).reshape(3, 4), "u": np.arange(5, dtype=np.int32)}}
    path = tmp_path / "predictions.h5"
    _hdf5.write_records(path, recs)
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"  # a real HDF5 file
    back = load_predictions(path)
    assert set(back) == set(recs)
    for n, rec in recs.items():
        assert set(back[n]) == set(rec)
        for k, v in rec.items():
            v = np.asarray(v)
            exp = v.astype(np.uint8) if v.dtype == np.bool_ else v
            assert back[n][k].dtype == exp.dtype and back[n][k].shape == exp.shape and np.array_equal(back[n][k], exp), (n, k)

    class Fake(torch.nn.Module):
        def forward(self, data):
            return {"keypoints": data["x"] * 3, "keypoint_scores": data["x"][..., 0], "descriptors": data["x"].repeat(1, 1, 4)}

    items = [{"name": [f"s/{i}.ppm"], "x": torch.rand((1, 5, 2), generator=g), "scales": torch.tensor([[0.5, 2.0]])}
             for i in range(3)]
    out = export_predictions(items, Fake(), tmp_path / "feats.h5", keys=["keypoints", "keypoint_scores", "descriptors"])
    assert open(out, "rb").read(4) == b"\x89HDF"
    loader = cache_loader.CacheLoader({"path": str(out), "add_data_path": False}).eval()
    got = loader({"name": [items[1]["name"][0]], "scales": items[1]["scales"]})
    assert torch.allclose(got["keypoints"][0], items[1]["x"][0] * 3)  # un-scaled on export, re-scaled on load
    assert torch.equal(got["keypoint_scores"][0], items[1]["x"][0, :, 0])


def test_disk_never_picks_up_a_third_party_network_implicitly(monkeypatch):
    """Config 5 (DISK): an importable `kornia` must NOT become the network by itself -- the product path never runs a
    third-party eager PyTorch network silently.  Without an explicit dense_fn the module stays uninitialised and its
    forward raises (reference wrapper: gluefactory/models/extractors/disk_kornia.py:24-47)."""
    import types

    from glue_factory_colon_amd import disk_kornia

    used = []
    fake = types.ModuleType("kornia")
    fake.feature = types.SimpleNamespace(DISK=lambda *a, **k: used.append("constructed"))
    monkeypatch.setitem(sys.modules, "kornia", fake)
    m = disk_kornia.DISK({"max_num_keypoints": 64})
    assert not used and not m.is_initialized()
    with pytest.raises((RuntimeError, AssertionError)):
        m({"image": torch.zeros(1, 3, 32, 32)})
    m2 = disk_kornia.DISK({"max_num_keypoints": 64}, dense_fn=lambda x: (x[:, :1], x))
    assert m2.is_initialized() and not used


def test_disk_network_state_dict_layout_and_checkpoint_file(tmp_path):
    """Config 5: the native DISK network carries kornia's parameter names (restated: oracle/disk_unet.py), so the module's
    state dict is `model.` + those keys (gluefactory/models/extractors/disk_kornia.py:24-28 keeps kornia's DISK as
    `self.model`), name-seeded weights and a kornia-style checkpoint file ({"extractor": state_dict}) load, and "depth"
    (a download) leaves the module un-initialised."""
    from glue_factory_colon_amd import disk_kornia, weights
    from oracle import disk_unet as ounet

    sd = weights.disk_state_dict(3)
    table = ounet.layer_table(128)
    assert [r[1:3] for r in table] == [(3, 16), (16, 32), (32, 64), (64, 64), (64, 64), (128, 64), (128, 64), (96, 64), (80, 129)]
    for prefix, cin, cout, gated in table:
        assert tuple(sd[prefix + ".3.weight"].shape) == (cout, cin, 5, 5) and tuple(sd[prefix + ".3.bias"].shape) == (cout,)
        assert (prefix + ".1.weight" in sd) == gated and (not gated or tuple(sd[prefix + ".1.weight"].shape) == (cin,))
    assert len(sd) == sum(3 if r[3] else 2 for r in table)
    m = disk_kornia.DISK({"weights": "synthetic:3", "max_num_keypoints": 32})
    assert m.is_initialized() and set(m.state_dict()) == {"model." + k for k in sd}
    assert all(torch.equal(m.state_dict()["model." + k], v) for k, v in sd.items())
    path = tmp_path / "disk_depth.ckpt"
    torch.save({"extractor": sd}, str(path))
    m2 = disk_kornia.DISK({"weights": str(path)})
    assert m2.is_initialized() and torch.equal(m2.state_dict()["model.unet.path_up.3.conv.3.bias"], sd["unet.path_up.3.conv.3.bias"])
    m3 = disk_kornia.DISK({})  # "depth"
    assert not m3.is_initialized()
    m3.load_state_dict(m.state_dict())
    assert m3.is_initialized()
    out = ounet.heatmap_and_dense_descriptors(sd, torch.rand(1, 3, 32, 48))
    assert out[0].shape == (1, 1, 32, 48) and out[1].shape == (1, 128, 32, 48)
    with pytest.raises(ValueError, match="divisible by 16"):
        ounet.unet(sd, torch.rand(1, 3, 30, 48))


def test_worker_replicas_share_weights_and_own_their_launch_state():
    """export_predictions(workers > 1): a replica is another view of the SAME parameters (nothing copied, so nothing to
    synchronise) with its own runners / workspaces / side streams; the module tree is cloned, not shared."""
    from glue_factory_colon_amd import export_predictions as ep
    from glue_factory_colon_amd.two_view_pipeline import TwoViewPipeline

    pipe = TwoViewPipeline({"extractor": {"name": "gluefactory_nonfree.superpoint", "weights": "synthetic",
                                          "max_num_keypoints": 64},
                            "matcher": {"name": "matchers.lightglue_pretrained", "weights": "synthetic"}}).eval()
    rep = ep._replicate(pipe)
    assert rep is not pipe and rep.extractor is not pipe.extractor and rep.matcher.net is not pipe.matcher.net
    for (n0, p0), (n1, p1) in zip(pipe.named_parameters(), rep.named_parameters()):
        assert n0 == n1 and p0 is p1
    assert rep.extractor._runner is not pipe.extractor._runner
    assert rep.matcher.net._ws is not pipe.matcher.net._ws
    assert rep.matcher.net._graphs is not pipe.matcher.net._graphs
    assert rep.extractor._packed is pipe.extractor._packed  # (None here: no GPU; shared once packed)
    assert rep.extractor.conf is pipe.extractor.conf and not rep.training
    # the combination that round 5 removed is refused before any GPU work
    import pytest as _pytest
    with _pytest.raises(ValueError, match="not combinable"):
        ep._export_loop(iter(()), pipe, "cuda", "*", [], None, False, 2, [], 2)


def test_bench_multiprocess_plumbing_rehearsal_world8():
    """The first 8-GPU run must not also be the first 8-rank run: bench.py --gpus 8 launched as the driver launches it,
    with --rehearse-cpu (gloo): rendezvous on 127.0.0.1, barriers, MAX all-reduce, the final gather of 8 x 32 records."""
    import json

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3",
           "--warmup", "1", "--pairs", "32", "--rehearse-cpu", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout + r.stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["pairs_gathered"] == 256  # BASELINE configs[3]: 256 pairs over 8 GPUs
    assert out["matches_per_rank"] == [10 + r for r in range(8)]
    # the line explains itself: every rank's own timings, gathered after the timed region (bench.py::per_rank_table)
    per = out["config"]["per_rank"]
    assert [row["rank"] for row in per] == list(range(8))
    for field in ("busy_s", "elapsed_s", "attention_avg_launch_ms", "stem_avg_launch_ms", "probe_tflops",
                  "probe_shader_clock_ghz", "matches_last_step"):
        assert all(field in row for row in per), field
    assert [row["matches_last_step"] for row in per] == [10 + r for r in range(8)]
    assert per[7]["busy_s"] > 2 * per[0]["busy_s"]  # the rehearsal makes rank r slower with r: visible per rank ...
    # ... while every rank waited for the slowest (each rank's clock starts when IT leaves the opening barrier: allow skew)
    assert all(row["elapsed_s"] >= per[7]["busy_s"] * 0.9 for row in per)
    summ = out["config"]["per_rank_summary"]
    assert summ["busy_s"][0] == per[0]["busy_s"] and summ["busy_s"][2] == per[7]["busy_s"]
    assert summ["busy_s"][0] <= summ["busy_s"][1] <= summ["busy_s"][2] and "probe_tflops" not in summ  # no GPU here


def test_export_predictions_sharded_world8_gloo_540_items(tmp_path):
    """The HPatches list's size (540 pairs: 108 sequences x 5) shared out over 8 ranks on gloo: every item is processed
    by exactly one rank (round-robin: rank r gets items r, r + 8, ...), rank 0 writes the records in LOADER order, and a
    second export in which rank 5 fails leaves every rank with an exception and no file."""
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os, torch, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from glue_factory_colon_amd import sharding\n"
        "from glue_factory_colon_amd.export_predictions import export_predictions, load_predictions\n"
        "rank, world, _ = sharding.init_from_env('gloo')\n"
        "seen = []\n"
        "class Fake(torch.nn.Module):\n"
        "    fail = False\n"
        "    def forward(self, data):\n"
        "        seen.append(int(data['idx']))\n"
        "        if self.fail and rank == 5 and len(seen) == 3:\n"
        "            raise ValueError('boom on rank 5')\n"
        "        k0 = data['view0']['x'] + rank * 0  # the result does not depend on the rank that computes it\n"
        "        return {'keypoints0': k0, 'matches0': (k0[..., 0] > 4).long() - 1, 'who': torch.full((1, 1), float(rank))}\n"
        "def item(i):\n"
        "    g = torch.Generator().manual_seed(i)\n"
        "    n = 3 + i % 4\n"
        "    return {'name': [f'seq{i // 5:03d}/{i % 5 + 2}.ppm'], 'idx': i,\n"
        "            'view0': {'x': torch.rand((1, n, 2), generator=g) * 9, 'scales': torch.tensor([[0.5 + 0.001 * i, 0.75]])}}\n"
        "items = [item(i) for i in range(540)]\n"
        "out = sys.argv[1]\n"
        "export_predictions(items, Fake(), out + '/p.npz', keys=['keypoints0', 'matches0', 'who'], pair_batch=4)\n"
        "assert seen == list(range(rank, 540, world)), (rank, seen[:5])\n"
        "if rank == 0:\n"
        "    a = load_predictions(out + '/p.npz')\n"
        "    assert list(a) == [it['name'][0] for it in items]\n"
        "    for i, it in enumerate(items):\n"
        "        r = a[it['name'][0]]\n"
        "        assert int(r['who'][0]) == i % world\n"
        "        assert np.array_equal(r['keypoints0'], (it['view0']['x'][0] * (1.0 / it['view0']['scales'])).numpy())\n"
        "    print('WORLD8_EXPORT_OK')\n"
        "torch.distributed.barrier()\n"
        "seen.clear()\n"
        "export_predictions(items, Fake(), out + '/g.npz', keys=['keypoints0', 'matches0', 'who'], pair_batch=5, shard_group=5)\n"
        "assert seen == [i for i in range(540) if (i // 5) % world == rank], (rank, seen[:12])  # whole sequences per rank\n"
        "if rank == 0:\n"
        "    b = load_predictions(out + '/g.npz')\n"
        "    assert list(b) == list(a) and all(np.array_equal(a[n]['keypoints0'], b[n]['keypoints0']) for n in a)\n"
        "torch.distributed.barrier()\n"
        "seen.clear(); Fake.fail = True\n"
        "try:\n"
        "    export_predictions(items, Fake(), out + '/q.npz', keys=['keypoints0'])\n"
        "    msg = 'NO_EXCEPTION'\n"
        "except ValueError as e:\n"
        "    msg = f'OWN {e}'\n"
        "except RuntimeError as e:\n"
        "    msg = f'TOLD {e}'\n"
        "open(out + f'/rank{rank}.txt', 'w').write(msg)\n"
        "assert not os.path.exists(out + '/q.npz')\n"
        "torch.distributed.barrier()\n")
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "WORLD8_EXPORT_OK" in r.stdout, r.stdout + r.stderr
    said = [(tmp_path / f"rank{i}.txt").read_text() for i in range(8)]
    assert said[5] == "OWN boom on rank 5", said
    assert all(s.startswith("TOLD") and "another rank failed" in s for i, s in enumerate(said) if i != 5), said

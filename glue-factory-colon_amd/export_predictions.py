"""Export loop -- counterpart of gluefactory/utils/export_predictions.py:21-91 for this package.

`for data in loader: pred = model(data)`; key filtering (`keys` / `optional_keys`, ValueError on a missing
key), un-scaling of key points to the original image resolution (`keypoints{i} * 1/view{i}.scales`), one record
per pair named `data["name"][0]` holding every exported key without its batch dimension.  The reference writes
HDF5 groups through h5py; an `.h5` output here is the same HDF5 layout, written through h5py when it is importable and
otherwise through the HDF5 C library itself (`_hdf5.py`, ctypes); other suffixes give one `.npz` archive with keys
`"<name>/<key>"` (`load_predictions` reads both).

MI355X additions.  The evaluation loop runs at batch 1 because the IMAGES of an HPatches-style list differ in size
(datasets/hpatches.py:60), and batch-1 kernels fill only part of the chip (conv4 at 60x80: 40 workgroups for 256 CUs).
* `pair_batch` = N > 1: N consecutive loader items are processed together (`model.forward_pairs`): the extractor once
  per distinct image shape among their 2N views, the matcher ONCE over all N pairs with their own key-point counts
  (LightGlue is image-size independent once the key points exist; gfc_lg_forward_ragged).  Records are those of the
  sequential loop (integers identical, floats within the batch-size dependence of the kernels' summation order) and
  are written in loader order.
* `workers` > 1 (with pair_batch 1): that many host threads, each with its own HIP stream and its own launch state
  (workspaces, side streams) around the SHARED read-only weights, process different pairs at the same time.  Not
  combinable with `pair_batch` > 1 (a pair batch already fills the chip; measured slower together: DESIGN.md 6).
"""
import queue
import threading
from pathlib import Path

import numpy as np
import torch


def _to_device(data, device):
    if isinstance(data, torch.Tensor):
        return data.to(device, non_blocking=True)
    if isinstance(data, dict):
        return {k: _to_device(v, device) for k, v in data.items()}
    if isinstance(data, (list, tuple)) and data and isinstance(data[0], torch.Tensor):
        return type(data)(_to_device(v, device) for v in data)
    return data


def _ensure_packed(model, device):
    """Every module of `model` that keeps packed device weights builds them now, on the calling thread's stream."""
    for m in model.modules():
        if hasattr(m, "ensure_packed") and getattr(m, "are_weights_initialized", True):
            m.ensure_packed(device)


def _replicate(model):
    """Another worker's view of `model`: the SAME parameters and the same packed device weights (read-only on the
    path), its own launch state -- runners / workspaces, side streams, captured graphs.  Nothing is copied on the
    device, so there is no copy whose completion a worker's stream would have to wait for; the caller packs the
    weights (`_ensure_packed`) and makes the worker streams wait on its own stream before the workers start."""
    from . import _native as nat
    from ._superpoint_common import SuperPointRunner

    def clone(m):
        c = m.__class__.__new__(m.__class__)
        c.__dict__ = dict(m.__dict__)
        c._modules = type(m._modules)((k, None if v is None else clone(v)) for k, v in m._modules.items())
        if "_runner" in c.__dict__:
            c._runner = SuperPointRunner()
        if "_ws" in c.__dict__:
            c._ws = nat.Workspace()
        if "_graphs" in c.__dict__:
            c._graphs = {}
        if "_view_lanes" in c.__dict__:
            c._view_lanes = None
        return c

    return clone(model)


def _process(model, data, keys, optional_keys, callback_fn, as_half):
    """One pair: forward, key filtering, un-scaling, host copy (export_predictions.py:36-85)."""
    return _record(model(data), data, keys, optional_keys, callback_fn, as_half)


def _process_batch(model, datas, keys, optional_keys, callback_fn, as_half, view_key=None):
    """`pair_batch` consecutive pairs through one `forward_pairs` call; one record per pair."""
    return _process_batch_async(model, datas, keys, optional_keys, callback_fn, as_half, view_key)()


class _HostStage:
    """Per-thread pinned staging buffers (one per dtype, grow-only) and the side stream the record copies run on."""

    def __init__(self):
        self.local = threading.local()

    def get(self, dtype, numel, device):
        st = self.local.__dict__
        if "stream" not in st:
            st["stream"], st["bufs"] = torch.cuda.Stream(device), {}
        # two alternating buffers per dtype: batch i's records are still being read while batch i + 1 is copied
        slot = st.setdefault("turn", {}).get(dtype, 0)
        st["turn"][dtype] = slot ^ 1
        buf = st["bufs"].get((dtype, slot))
        if buf is None or buf.numel() < numel:
            buf = torch.empty(int(numel * 1.25) + 16, dtype=dtype).pin_memory()
            st["bufs"][(dtype, slot)] = buf
        return st["stream"], buf


_HOST_STAGE = _HostStage()


def _process_batch_async(model, datas, keys, optional_keys, callback_fn, as_half, view_key=None):
    """Forward + record building of one pair batch; returns a function that completes the host copy and yields the
    records.  ONE device-to-host copy per dtype for the whole batch (instead of one synchronising copy per pair and
    key), issued on a side stream into pinned memory: the caller may queue the next batch's forward before it calls the
    function, so that slicing this batch's records overlaps the next batch's kernels."""
    if hasattr(model, "forward_pairs"):
        if view_key is not None:
            preds = model.forward_pairs(datas, view_keys=[(view_key(d, 0), view_key(d, 1)) for d in datas])
        else:
            preds = model.forward_pairs(datas)
    else:
        preds = [model(d) for d in datas]
    preds = [_record(p, d, keys, optional_keys, callback_fn, as_half, to_host=False) for p, d in zip(preds, datas)]
    by_dtype = {}
    for i, p in enumerate(preds):
        for k, v in p.items():
            by_dtype.setdefault(v.dtype, []).append((i, k, tuple(v.shape[1:]), v[0].reshape(-1)))
    staged = []
    for dtype, entries in by_dtype.items():
        flat = torch.cat([t for *_, t in entries])
        if flat.device.type == "cuda":
            side, host = _HOST_STAGE.get(dtype, flat.numel(), flat.device)
            side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(side):
                host[:flat.numel()].copy_(flat, non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
            flat.record_stream(side)
            staged.append((entries, host, done))
        else:
            staged.append((entries, flat, None))
    order = [list(p) for p in preds]

    def finish():
        recs = [dict() for _ in order]
        for entries, host, done in staged:
            if done is not None:
                done.synchronize()
            arr = host.numpy()
            off = 0
            for i, k, shape, t in entries:
                n = t.numel()
                recs[i][k] = arr[off:off + n].reshape(shape).copy()
                off += n
        out = [{k: r[k] for k in ks} for ks, r in zip(order, recs)]  # key order of the prediction
        if as_half:
            out = [{k: (v.astype(np.float16) if v.dtype == np.float32 else v) for k, v in r.items()} for r in out]
        return out

    return finish


def _record(pred, data, keys, optional_keys, callback_fn, as_half, to_host=True):
    if callback_fn is not None:
        pred = {**callback_fn(pred, data), **pred}
    if keys != "*":
        missing = set(keys) - set(pred.keys())
        if missing:
            raise ValueError(f"Missing key {missing}")
        pred = {k: v for k, v in pred.items() if k in list(keys) + optional_keys}
    assert len(pred) > 0
    for k in list(pred.keys()):  # back to the resolution of the original image
        if k.startswith("keypoints"):
            idx = k.replace("keypoints", "")
            scales = 1.0 / (data["scales"] if len(idx) == 0 else data[f"view{idx}"]["scales"])
            pred[k] = pred[k] * scales[None]
    if not to_host:
        return pred
    rec = {k: v[0].cpu().numpy() for k, v in pred.items()}  # .cpu() waits for this thread's stream only
    if as_half:
        rec = {k: (v.astype(np.float16) if v.dtype == np.float32 else v) for k, v in rec.items()}
    return rec


def _sharded(loader, rank, world, group=1):
    """(index, item) of this rank's round-robin share of the loader (sharding.round_robin_shard: pairs of an
    HPatches-style list differ in size, so consecutive pairs go to different ranks).  A loader that can skip work
    itself may offer `shard(rank, world)` yielding (index, item); a sequential batch-1 torch DataLoader (what the
    reference's evaluation builds, datasets/base_dataset.py:175-186) is re-created over this rank's Subset of its
    dataset, so that each rank LOADS only its own items; otherwise the other ranks' items are drawn and dropped."""
    from torch.utils.data import DataLoader, SequentialSampler, Subset
    if world <= 1:
        yield from enumerate(loader)
    elif hasattr(loader, "shard"):
        yield from (loader.shard(rank, world, group) if group > 1 else loader.shard(rank, world))
    elif (isinstance(loader, DataLoader) and loader.batch_size == 1 and isinstance(loader.sampler, SequentialSampler)
          and hasattr(loader.dataset, "__len__") and hasattr(loader.dataset, "__getitem__")):
        from .sharding import round_robin_shard
        idx = list(round_robin_shard(len(loader.dataset), rank, world, group))
        extra = {"timeout": loader.timeout, "generator": loader.generator}
        if loader.num_workers > 0:  # only valid with worker processes
            extra.update(prefetch_factor=loader.prefetch_factor, persistent_workers=loader.persistent_workers,
                         multiprocessing_context=loader.multiprocessing_context)
        sub = DataLoader(Subset(loader.dataset, idx), batch_size=1, shuffle=False, num_workers=loader.num_workers,
                         collate_fn=loader.collate_fn, pin_memory=loader.pin_memory,
                         worker_init_fn=loader.worker_init_fn, **extra)
        yield from zip(idx, sub)
    elif hasattr(loader, "__getitem__") and hasattr(loader, "__len__"):
        from .sharding import round_robin_shard
        for i in round_robin_shard(len(loader), rank, world, group):
            yield i, loader[i]
    else:
        for i, item in enumerate(loader):
            if (i // max(group, 1)) % world == rank:
                yield i, item


def _gather_to_rank0(local, failed_here, rank, world, device):
    """the ONE data collective of the sharded export -- every rank's records travel to rank 0 as one block
    of bytes, padded to the largest block (the npz container inside keeps each pair's arrays with their own shapes
    and dtypes: an HPatches pair is ~40 KB, 540 pairs ~22 MB over 8 ranks).  A one-integer all-gather in front of it
    carries the block sizes and doubles as the failure flag (-1): a rank that failed must not leave the others
    waiting in a collective.  Returns (the ranks' raw blocks on rank 0 / None elsewhere, any_rank_failed)."""
    import io

    import torch.distributed as dist
    dev = device if dist.get_backend() == "nccl" else "cpu"
    payload = b""
    if not failed_here:
        buf = io.BytesIO()
        np.savez(buf, **{f"{idx}|{name}/{k}": v for idx, name, rec in local for k, v in rec.items()})
        payload = buf.getvalue()
    mine = torch.tensor([-1 if failed_here else len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [int(t.item()) for t in sizes]
    if min(sizes) < 0:
        return None, True
    block = torch.zeros(max(max(sizes), 1), dtype=torch.uint8)
    if payload:
        block[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    block = block.to(dev)
    out = [torch.empty_like(block) for _ in range(world)] if rank == 0 else None
    dist.gather(block, out, dst=0)
    if rank != 0:
        return None, False
    # raw blocks: decoded by the caller inside the try that feeds the closing failure flag
    return [out[r][:sizes[r]].cpu().numpy().tobytes() for r in range(world)], False


def _decode_blocks(blocks):
    """[(index, name, record)] from the ranks' npz blocks."""
    import io
    entries = []
    for raw in blocks:
        with np.load(io.BytesIO(raw), allow_pickle=False) as z:
            recs = {}
            for full in z.files:
                head, key = full.rsplit("/", 1)
                recs.setdefault(head, {})[key] = z[full]
        for head, rec in recs.items():
            idx, name = head.split("|", 1)
            entries.append((int(idx), name, rec))
    return entries


@torch.no_grad()
def export_predictions(loader, model, output_file, as_half=False, keys="*", callback_fn=None, optional_keys=(),
                       workers=1, rank=None, world=None, pair_batch=1, view_key=None, shard_group=1):
    """pair_batch: number of consecutive pairs processed by one `model.forward_pairs` call (module docstring).
    view_key (with pair_batch > 1): `view_key(item, i) -> hashable or None` names the image of view i of a loader item;
    views of one pair batch with equal non-None names are extracted ONCE (TwoViewPipeline.forward_pairs).  For the
    HPatches list (datasets/hpatches.py:98-112: view 0 of every pair of a sequence is the sequence's image 1):
    `view_key=lambda item, i: (item["scene"][0], 1) if i == 0 else None`.  Records are unchanged.
    shard_group: under torch.distributed the list is dealt out round-robin `shard_group` consecutive items at a time
    (5 keeps an HPatches sequence, whose pairs share their view 0, together on one rank: what `view_key` needs).
    rank / world (default: the torch.distributed process group, if one is initialised): the pair list is shared
    out round-robin over the ranks (one process per GPU, no data-path collective); at the end ONE gather brings every
    rank's records to rank 0 (RCCL over xGMI: one direct peer write per rank), which writes the single
    prediction file the evaluation reads (records in loader order, as the single-process loop writes them) -- no
    part files, no shared file system between the ranks.  All ranks return `output_file`."""
    assert keys == "*" or isinstance(keys, (tuple, list))
    optional_keys = list(optional_keys)
    output_file = Path(output_file)
    output_file.parent.mkdir(exist_ok=True, parents=True)
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if world > 1 else 0
    device = "cuda" if torch.cuda.is_available() else "cpu"
    model = model.to(device).eval()
    if world > 1:
        failure = None
        local = []
        try:
            _export_loop(_sharded(loader, rank, world, int(shard_group)), model, device, keys, optional_keys, callback_fn, as_half,
                         workers, local, pair_batch, view_key)
        except Exception as e:  # noqa: BLE001 -- re-raised below, after the other ranks have been told
            failure = e
        blocks, failed = _gather_to_rank0(local, failure is not None, rank, world, device)
        if failed:
            if failure is not None:
                raise failure
            raise RuntimeError("export_predictions: another rank failed; nothing was written")
        try:
            if rank == 0:  # records in loader order, first name wins (as the single-process loop)
                records = {}
                for _, name, rec in sorted(_decode_blocks(blocks), key=lambda e: e[0]):
                    if name not in records:
                        records[name] = rec
                _write(output_file, records)
        except Exception as e:  # noqa: BLE001
            failure = e
        failed = _any_rank_failed(failure is not None, device)  # closing barrier: the file exists when any rank returns
        if failure is not None:
            raise failure
        if failed:
            raise RuntimeError("export_predictions: writing the prediction file failed on rank 0")
        return output_file
    local = []
    _export_loop(enumerate(loader), model, device, keys, optional_keys, callback_fn, as_half, workers, local,
                 pair_batch, view_key)
    records = {}
    for _, name, rec in sorted(local, key=lambda e: e[0]):
        if name not in records:  # like the reference: a duplicate group name is skipped
            records[name] = rec
    _write(output_file, records)
    return output_file


def _any_rank_failed(failed_here: bool, device) -> bool:
    """One all-reduce (MAX) of a per-rank failure flag; doubles as the barrier between the export and merge phases."""
    import torch.distributed as dist
    backend = dist.get_backend()
    flag = torch.tensor([1 if failed_here else 0], dtype=torch.int32,
                        device=device if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return bool(flag.item())


def _batches(indexed, n):
    """Lists of up to n consecutive (index, item) entries."""
    chunk = []
    for entry in indexed:
        chunk.append(entry)
        if len(chunk) == n:
            yield chunk
            chunk = []
    if chunk:
        yield chunk


def _export_loop(indexed, model, device, keys, optional_keys, callback_fn, as_half, workers, out, pair_batch=1,
                 view_key=None):
    """Process (index, item) pairs; appends (index, name, record) to `out`."""
    pair_batch = max(1, int(pair_batch or 1))
    if workers > 1 and pair_batch > 1 and device != "cpu":
        raise ValueError("export_predictions: workers > 1 and pair_batch > 1 are not combinable (a pair batch fills the "
                         "chip by itself; use pair_batch alone)")

    def run_chunk(replica, chunk):
        """-> [(index, name, record)] of up to pair_batch consecutive items"""
        datas = [_to_device(d, device) for _, d in chunk]
        names = [d.get("name", [None])[0] for d in datas]
        if len(datas) == 1:
            recs = [_process(replica, datas[0], keys, optional_keys, callback_fn, as_half)]
        else:
            recs = _process_batch(replica, datas, keys, optional_keys, callback_fn, as_half, view_key)
        return [(idx, name, rec) for (idx, _), name, rec in zip(chunk, names, recs)]

    if workers <= 1 or device == "cpu":
        if pair_batch > 1 and device != "cpu":
            # software pipeline of depth one: batch i's records are cut on the host while batch i + 1's kernels run
            pending = None
            for chunk in _batches(indexed, pair_batch):
                datas = [_to_device(d, device) for _, d in chunk]
                names = [d.get("name", [None])[0] for d in datas]
                fin = _process_batch_async(model, datas, keys, optional_keys, callback_fn, as_half, view_key)
                if pending is not None:
                    out.extend((idx, name, rec) for (idx, _), name, rec in zip(pending[0], pending[1], pending[2]()))
                pending = (chunk, names, fin)
            if pending is not None:
                out.extend((idx, name, rec) for (idx, _), name, rec in zip(pending[0], pending[1], pending[2]()))
            return
        for chunk in _batches(indexed, pair_batch):
            out.extend(run_chunk(model, chunk))
        return

    # ---- `workers` pairs in flight: one thread + one HIP stream + its own launch state each, shared weights ----
    dev = torch.device("cuda", torch.cuda.current_device())
    main = torch.cuda.current_stream(dev)
    _ensure_packed(model, dev)  # on the caller's stream, which every worker stream waits on before its first launch
    replicas = [_replicate(model) for _ in range(workers)]  # the model's own launch state stays with the caller's stream
    streams = [torch.cuda.Stream(dev) for _ in replicas]
    for st in streams:
        st.wait_stream(main)
    tasks: "queue.Queue" = queue.Queue(maxsize=2 * workers)
    results, errors = [], []

    def device_tensors(obj):
        if torch.is_tensor(obj):
            if obj.is_cuda:
                yield obj
        elif isinstance(obj, dict):
            for v in obj.values():
                yield from device_tensors(v)
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                yield from device_tensors(v)

    def run(replica, stream):
        with torch.no_grad(), torch.cuda.stream(stream):
            while True:
                task = tasks.get()
                if task is None:
                    return
                chunk, ready = task
                try:
                    if ready is not None:
                        # the items arrived ON THE DEVICE (image_preprocessor.HostImageFeeder: H2D copy + resize kernel
                        # queued on the producing thread's stream): order this stream after the producer, and tell the
                        # caching allocator that this stream uses the tensors, so their memory is not recycled for the
                        # producer's next item while a kernel here still reads it
                        stream.wait_event(ready)
                        for _, item in chunk:
                            for t in device_tensors(item):
                                t.record_stream(stream)
                    results.extend(run_chunk(replica, chunk))  # list.extend is atomic under the GIL
                except Exception as e:  # noqa: BLE001 -- re-raised in the caller's thread
                    errors.append(e)

    threads = [threading.Thread(target=run, args=(r, st), daemon=True) for r, st in zip(replicas, streams)]
    for t in threads:
        t.start()
    for chunk in _batches(indexed, pair_batch):  # the loader is iterated HERE, on the caller's thread and stream
        if errors:
            break
        ready = None
        if any(True for _, item in chunk for _ in device_tensors(item)):
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))  # after everything the loader queued for these items
        tasks.put((chunk, ready))
    for _ in threads:
        tasks.put(None)
    for t in threads:
        t.join()
    for st in streams:  # every record was copied to the host by its worker; the caller's stream continues after them
        main.wait_stream(st)
    if errors:
        raise errors[0]
    out.extend(sorted(results, key=lambda e: e[0]))


def _write(path: Path, records: dict):
    """`.h5` / `.hdf5`: the reference's HDF5 layout (group per record name, dataset per key) through h5py when it is
    importable, else through the HDF5 C library bound in `_hdf5.py`; any other suffix (or no HDF5 library at all):
    one `.npz` archive with keys "<name>/<key>"."""
    if path.suffix in (".h5", ".hdf5"):
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            with h5py.File(str(path), "w") as f:
                for name, rec in records.items():
                    grp = f.create_group(name)
                    for k, v in rec.items():
                        grp.create_dataset(k, data=v)
            return
        from . import _hdf5
        if _hdf5.available():
            _hdf5.write_records(path, records)
            return
    flat = {f"{name}/{k}": v for name, rec in records.items() for k, v in rec.items()}
    with open(path, "wb") as fh:
        np.savez(fh, **flat)


def load_predictions(path):
    """{name: {key: ndarray}} from either container (HDF5: names in alphabetical order, as HDF5 iterates them)."""
    path = Path(path)
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic == b"\x89HDF":
        try:
            import h5py
        except ImportError:
            from . import _hdf5
            return _hdf5.read_records(path)
        with h5py.File(str(path), "r") as f:
            out = {}

            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    grp, key = name.rsplit("/", 1)
                    out.setdefault(grp, {})[key] = obj[()]

            f.visititems(visit)
            return out
    out = {}
    with np.load(path) as z:
        for full in z.files:
            name, key = full.rsplit("/", 1)
            out.setdefault(name, {})[key] = z[full]
    return out

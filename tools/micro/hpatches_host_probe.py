"""Where the time of config 3 from files goes: file read into pinned memory (raw_loader), copy + resize (feeder),
extraction + matching + record handling + file write (export_predictions), for several reader-thread counts and torch
thread-pool sizes."""
import os, sys, time, tempfile, shutil
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glue_factory_colon_amd import eval_hpatches, hpatches, synthetic
from glue_factory_colon_amd.export_predictions import export_predictions

N = int(sys.argv[1]) if len(sys.argv) > 1 else 240
root = tempfile.mkdtemp(prefix="gfc_hp_")
try:
    raw = synthetic.hpatches_like_host_images(N, seed=7000, pin=False, shared_view0=True)
    for i, it in enumerate(raw):
        d = os.path.join(root, "hp", "v_" + it["scene"]); os.makedirs(d, exist_ok=True)
        for name, img in ((("1.ppm", it["view0"]["image"]),) if i % 5 == 0 else ()) + ((f"{i % 5 + 2}.ppm", it["view1"]["image"]),):
            a = img.numpy(); h, w = a.shape[:2]
            open(os.path.join(d, name), "wb").write(b"P6\n" + f"{w} {h}\n255\n".encode() + a.tobytes())
        open(os.path.join(d, f"H_1_{i % 5 + 2}"), "w").write("1 0 0\n0 1 0\n0 0 1\n")
    del raw
    ds = hpatches.HPatches({"data_dir": os.path.join(root, "hp"), "preprocessing": {"resize": 480, "side": "short"}})
    model = eval_hpatches.build_model("synthetic", "synthetic", official=True).cuda()
    keys = eval_hpatches.EXPORT_KEYS
    for threads in (torch.get_num_threads(),):
        torch.set_num_threads(threads)
        print(f"== torch threads {threads}", flush=True)
        for w in (0, 2):
            for rep in range(2):
                t0 = time.perf_counter(); n = sum(1 for _ in ds.raw_loader(None, w)); dt = time.perf_counter() - t0
            print(f"raw_loader        workers {w}: {n / dt:7.1f} pairs/s", flush=True)
            for rep in range(2):
                t0 = time.perf_counter(); n = sum(1 for _ in ds.feeder(num_workers=w)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"feeder            workers {w}: {n / dt:7.1f} pairs/s", flush=True)
            import cProfile, pstats
            if w == 2 and "--profile" in sys.argv:
                pr = cProfile.Profile(); pr.enable()
                export_predictions(ds.feeder(num_workers=w), model, os.path.join(root, "p.npz"), keys=keys, pair_batch=32, view_key=ds.view_key)
                torch.cuda.synchronize(); pr.disable()
                pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
            for suffix in ("npz", "h5"):
                for rep in range(2):
                    t0 = time.perf_counter()
                    export_predictions(ds.feeder(num_workers=w), model, os.path.join(root, f"p.{suffix}"), keys=keys, pair_batch=32, view_key=ds.view_key)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                print(f"export -> .{suffix:3s}     workers {w}: {N / dt:7.1f} pairs/s", flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)

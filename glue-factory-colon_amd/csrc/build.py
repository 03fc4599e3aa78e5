"""Build libgfc_amd.so (all HIP kernels + the C ABI) for gfx950 with hipcc.

    python glue-factory-colon_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU; the .so is written next to the Python package
(in-tree, git-ignored) so that it travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["runtime.hip", "conv.hip", "conv_split.hip", "conv_wino.hip", "gemm.hip", "attention.hip", "attention_split.hip", "sp_detect.hip", "disk_detect.hip", "lg_misc.hip", "eval_metrics.hip", "preprocess.hip", "api.hip"]
HEADERS = ["common.h", "runtime.h", os.path.join(PKG, "..", "include", "gfc_amd.h")]
LIB = os.path.join(PKG, "libgfc_amd.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src, *hdrs]):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for (src, obj), r in ex.map(compile_one, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    print(r.stderr, file=sys.stderr)
                if r.returncode:
                    raise RuntimeError(f"hipcc failed on {src}")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr, file=sys.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

"""Build libgfc_amd.so (all HIP kernels + the C ABI) for gfx950 with hipcc.

    python glue-factory-colon_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU; the .so is written next to the Python package
(in-tree, git-ignored) so that it travels to the GPU box with the repository snapshot.

What is rebuilt is decided by CONTENT, not by file times: every object file carries the sha256 of
(compiler flags, its source, every header) in `build/<unit>.o.sha256`, and the library carries the
hash of the whole source set -- both in `build/lib.sha256` and inside the binary: `gfc_version()`
ends in `src <hash>`, so a process can tell that the library it loaded was built from the sources
beside it (`source_hash()` here; __graft_entry__.build() checks exactly that).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
SOURCES = ["runtime.hip", "conv.hip", "conv_wino.hip", "conv_wino43.hip", "gemm.hip", "attention.hip", "sp_detect.hip", "sp_heads.hip", "disk_detect.hip", "disk_unet.hip", "lg_misc.hip", "eval_metrics.hip", "preprocess.hip", "api.hip"]
HEADERS = ["common.h", "runtime.h", os.path.join(PKG, "..", "include", "gfc_amd.h")]
LIB = os.path.join(PKG, "libgfc_amd.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
VERSION_UNIT = "api.hip"  # compiled with -DGFC_SOURCE_HASH="<hash>": gfc_version() reports it
# Second builds of the same library with other arithmetic, for the tests that bound a deliberate approximation
# (GFC_AMD_LIB=<path> selects one): name -> (extra flags, the translation units those flags change; every other
# object is shared with the default build).
VARIANTS = {
    # GELU through the exact erf chain instead of Abramowitz & Stegun 7.1.26 (common.h: gfc_gelu)
    "exact_erf": (["-DGFC_EXACT_ERF=1"], ["gemm.hip", "lg_misc.hip"]),
}


def variant_lib(name):
    return os.path.join(PKG, f"libgfc_amd_{name}.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def _headers():
    return [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]


def source_hash():
    """sha256 (first 12 hex digits) over the compiler flags, every source and every header, in a fixed order."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in [os.path.join(HERE, s) for s in SOURCES] + _headers():
        h.update(os.path.basename(path).encode() + b"\0" + _read(path) + b"\0")
    return h.hexdigest()[:12]


def _unit_hash(src, extra_flags):
    h = hashlib.sha256(" ".join(FLAGS + extra_flags).encode())
    for path in [src] + _headers():
        h.update(os.path.basename(path).encode() + b"\0" + _read(path) + b"\0")
    return h.hexdigest()


def _stamp_matches(stamp, value):
    try:
        return _read(stamp).decode().strip() == value
    except OSError:
        return False


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    whole = source_hash()
    jobs = []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        extra = [f'-DGFC_SOURCE_HASH="{whole}"'] if s == VERSION_UNIT else []
        want = _unit_hash(src, extra)
        if force or not os.path.exists(obj) or not _stamp_matches(obj + ".sha256", want):
            jobs.append((src, obj, extra, want))

    def compile_one(job):
        src, obj, extra, _ = job
        cmd = [hipcc, *FLAGS, *extra, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for (src, obj, _, want), r in ex.map(compile_one, jobs):
                if verbose and (r.stderr.strip() or r.returncode):
                    print(r.stderr, file=sys.stderr)
                if r.returncode:
                    raise RuntimeError(f"hipcc failed on {src}")
                with open(obj + ".sha256", "w") as f:
                    f.write(want)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    lib_stamp = os.path.join(objdir, "lib.sha256")
    relink = force or bool(jobs) or not os.path.exists(LIB) or not _stamp_matches(lib_stamp, whole)
    if relink:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stderr, file=sys.stderr)
            raise RuntimeError("link failed")
        with open(lib_stamp, "w") as f:
            f.write(whole)
    if verbose:
        print(f"gfc build: source hash {whole}; compiled {len(jobs)} of {len(SOURCES)} translation units "
              f"({'forced' if force else 'content hash changed or object missing'}), "
              f"{'linked' if relink else 'library up to date'}", file=sys.stderr)
    for name, (vflags, vunits) in VARIANTS.items():
        vdir = os.path.join(objdir, name)
        os.makedirs(vdir, exist_ok=True)
        vjobs = []
        for u in vunits:
            src, obj = os.path.join(HERE, u), os.path.join(vdir, u.replace(".hip", ".o"))
            want = _unit_hash(src, vflags)
            if force or not os.path.exists(obj) or not _stamp_matches(obj + ".sha256", want):
                vjobs.append((src, obj, vflags, want))
        for (src, obj, _, want), r in map(compile_one, vjobs):
            if r.returncode:
                print(r.stderr, file=sys.stderr)
                raise RuntimeError(f"hipcc failed on {src} ({name})")
            with open(obj + ".sha256", "w") as f:
                f.write(want)
        vobjs = [os.path.join(vdir if s in vunits else objdir, s.replace(".hip", ".o")) for s in SOURCES]
        vstamp = os.path.join(vdir, "lib.sha256")
        if relink or vjobs or not os.path.exists(variant_lib(name)) or not _stamp_matches(vstamp, whole):
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", variant_lib(name), *vobjs],
                               capture_output=True, text=True)
            if r.returncode:
                print(r.stderr, file=sys.stderr)
                raise RuntimeError(f"link failed ({name})")
            with open(vstamp, "w") as f:
                f.write(whole)
        if verbose:
            print(f"gfc build: variant {name}: compiled {len(vjobs)} of {len(vunits)} units", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

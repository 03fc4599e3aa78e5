"""Static issue census of the MFMA kernels' main loops (no GPU needed).

For each kernel the tool compiles its translation unit to gfx950 assembly (`hipcc --cuda-device-only -S`, the flags of
csrc/build.py), finds the loops of the kernel (a backward branch to an earlier label), and counts per loop the
instructions by class.  With the co-issue costs measured on the hardware (profiles/r05_mfma_valu_hybrid.txt: cycles of
matrix-pipe time one instruction of a class costs a SIMD that is otherwise saturated with v_mfma_f32_32x32x2_f32 from two
waves) it predicts the ceiling of the matrix pipe's utilisation for that loop,

    ceiling = MFMA cycles / (MFMA cycles + sum over VALU / LDS instructions of their co-issue cost),

to set beside the measured MfmaUtil (rocprofv3 --pmc, profiles/r05_pmc_summary.json).  A kernel AT its ceiling is
issue-bound (cut instructions); one far BELOW it is stall-bound (waits: the `s_waitcnt` / `s_barrier` census says where).

Limits of a STATIC count: a loop body holds blocks that most iterations skip (attention: the key-mask block of the last
tile, 212 of 559 vector instructions; the stem: three variants of the conv1a store, one of which runs), and the loops
the compiler rotates overlap.  The EXECUTED mix per kernel comes from the hardware counters instead
(tools/pmc_instmix.py, profiles/r06_instmix.json); this tool shows where in the code the instructions sit.

    python tools/issue_census.py [--out profiles/r06_issue_census.txt]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "glue-factory-colon_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S"]

# kernel (substring of the demangled name) -> (source, measured MfmaUtil % of profiles/r05_pmc_summary.json)
KERNELS = [
    ("stem_wino43_kernel", "conv_wino43.hip", 48.7),
    ("conv3x3_wino_kernel<true, true", "conv_wino.hip", 70.5),
    ("conv3x3_wino_kernel<false, false", "conv_wino.hip", 70.5),
    ("gemm_rows512_ln_gelu_kernel<2, true", "gemm.hip", 76.2),
    ("gemm_nt_kernel<2, 2, 16, 2", "gemm.hip", 73.0),
    ("attention_kernel<2, 4", "attention.hip", 82.3),
]

# cycles of matrix-pipe time per instruction and wave (r05_mfma_valu_hybrid.txt: (cycles with 8 per MFMA - 512) / 64)
MFMA_CYCLES = {"32x32x2": 64, "16x16x4": 32, "32x32x1": 64, "16x16x1": 32, "4x4x1": 8}
COST = collections.OrderedDict([
    ("valu_pk_fma", 6.8),    # v_pk_fma_f32, v_pk_mul_f32 (two passes)
    ("valu_pk_add", 5.0),    # v_pk_add_f32
    ("valu_trans", 9.8),     # v_exp / v_log / v_rcp / v_rsq / v_sqrt
    ("valu_sel", 4.9),       # v_max / v_min / v_cndmask / v_cmp
    ("valu_mov", 3.9),       # v_mov / v_accvgpr moves / v_readlane etc.
    ("valu_simple", 3.2),    # v_fma / v_add / v_mul / v_and / shifts / integer
    # LDS: the ISSUE cost of one instruction (ds_read_b32 measured 1.2).  The 8.1 measured for ds_read_b128 at 8 per MFMA is
    # LDS BANDWIDTH (64 lanes x 16 B per instruction against ~256 B / clock / CU), not issue: wide reads at the rates of
    # these kernels (<= 0.5 per MFMA) are far from it, so they are priced at the issue cost too.
    ("lds_b128", 1.2),
    ("lds_b64", 1.2),
    ("lds_b32", 1.2),
    ("vmem", 0.0),           # global / buffer loads and stores: issue only (not measured as a matrix-pipe cost)
    ("salu", 0.0),           # scalar: other issue port (s_nop measured 0)
])


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_pk_fma", "v_pk_mul")):
        return "valu_pk_fma"
    if op.startswith("v_pk_"):
        return "valu_pk_add"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith(("v_max", "v_min", "v_cndmask", "v_cmp", "v_med3")):
        return "valu_sel"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap", "v_permlane", "v_bfe",
                      "v_perm")):
        return "valu_mov"
    if op.startswith("v_"):
        return "valu_simple"
    if op.startswith("ds_"):
        if "b128" in op:
            return "lds_b128"
        if "b64" in op or "b96" in op:
            return "lds_b64"
        return "lds_b32"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def kernels_of(asm):
    """{mangled name: [lines]} of every function body of an assembly file."""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is not None:
            cur.append(line)
            if line.strip().startswith(".end_amdhsa_kernel") or re.match(r"^\s*\.size\s", line):
                cur, name = None, None
    return out


def loops_of(lines):
    """[(first line index, last line index)] of every loop: a branch to a label that stands earlier in the function."""
    label_at = {}
    for i, line in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for i, line in enumerate(lines):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", line)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            loops.append((label_at[m.group(1)], i))
    return loops


def census(lines):
    counts = collections.Counter()
    mfma_cycles = 0
    for line in lines:
        m = re.match(r"^\s+([a-z_0-9]+)", line)
        if not m or line.lstrip().startswith((".", ";")):
            continue
        op = m.group(1)
        cls = classify(op)
        counts[cls] += 1
        if cls == "mfma":
            shape = re.search(r"_(\d+x\d+x\d+)", op)
            mfma_cycles += MFMA_CYCLES.get(shape.group(1) if shape else "", 64)
        if cls == "waitcnt":
            counts["waitcnt_lgkm" if "lgkmcnt" in line else "waitcnt_vm" if "vmcnt" in line else "waitcnt_other"] += 1
    return counts, mfma_cycles


def main():
    out_path = os.path.join(ROOT, "profiles", "r06_issue_census.txt")
    if "--out" in sys.argv:
        out_path = sys.argv[sys.argv.index("--out") + 1]
    asm_cache = {}
    report = [__doc__.split("\n\n")[0], "",
              "co-issue cost per instruction and wave, cycles of matrix-pipe time (profiles/r05_mfma_valu_hybrid.txt):",
              "  " + ", ".join(f"{k} {v}" for k, v in COST.items()) + "; v_mfma_f32_32x32x2_f32 64, 16x16x4 32", ""]
    with tempfile.TemporaryDirectory() as tmp:
        for want, src, measured in KERNELS:
            if src not in asm_cache:
                path = os.path.join(tmp, src.replace(".hip", ".s"))
                r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", CSRC, "-o", path, os.path.join(CSRC, src)],
                                   capture_output=True, text=True)
                if r.returncode:
                    raise SystemExit(r.stderr[-2000:])
                with open(path) as f:
                    ks = kernels_of(f.read())
                asm_cache[src] = (ks, demangle(list(ks)))
            ks, names = asm_cache[src]
            hits = [k for k in ks if want in names.get(k, "")]
            if not hits:
                report.append(f"== {want}: not found in {src}")
                continue
            lines = ks[hits[0]]
            total, _ = census(lines)
            vgpr = next((ln.split()[-1] for ln in lines if ".amdhsa_next_free_vgpr" in ln), "?")
            agpr = next((ln.split()[-1] for ln in lines if ".amdhsa_accum_offset" in ln), "?")
            report.append(f"== {names[hits[0]].split('(')[0]}   [{src}; next_free_vgpr {vgpr}, accum_offset {agpr}; whole kernel: "
                          f"{total['mfma']} MFMA, {sum(v for k, v in total.items() if k.startswith('valu'))} VALU, "
                          f"{sum(v for k, v in total.items() if k.startswith('lds'))} LDS, {total['vmem']} VMEM, "
                          f"{total['waitcnt']} s_waitcnt, {total['barrier']} s_barrier]")
            loops = [(a, b, *census(lines[a:b + 1])) for a, b in loops_of(lines)]
            loops = [lp for lp in loops if lp[2]["mfma"] > 0]
            # innermost first: a loop that contains another loop with MFMAs is listed after it, marked "outer"
            loops.sort(key=lambda lp: lp[1] - lp[0])
            for a, b, c, mc in loops:
                inner = [lp for lp in loops if lp[0] > a and lp[1] < b]
                extra = sum(COST.get(k, 0.0) * v for k, v in c.items())
                ceiling = mc / (mc + extra) if mc else 0.0
                cls = ", ".join(f"{k} {c[k]}" for k in COST if c[k])
                report.append(f"   loop lines {a}-{b}{' (outer: contains ' + str(len(inner)) + ' MFMA loop(s))' if inner else ''}: "
                              f"{c['mfma']} MFMA = {mc} cycles; {cls}; s_waitcnt {c['waitcnt']} (lgkm {c['waitcnt_lgkm']}, vm "
                              f"{c['waitcnt_vm']}), s_barrier {c['barrier']}")
                report.append(f"      co-issue cost {extra:.0f} cycles -> predicted ceiling {100 * ceiling:.1f} %  "
                              f"(measured MfmaUtil of the kernel {measured} %)")
            report.append("")
    text = "\n".join(report)
    with open(out_path, "w") as f:
        f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()

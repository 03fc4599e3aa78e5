"""Dynamic instruction mix and wait cycles per kernel from rocprofv3 --pmc passes of `bench.py` (tools/profile_r06.sh):
the EXECUTED counterpart of tools/issue_census.py's static loop census.

    python tools/pmc_instmix.py <pass dir> [<pass dir> ...] > profiles/r06_instmix.json

Every directory is one `rocprofv3 --pmc <counters>` run; counters are averaged per launch of a kernel (summed over the
chip by the profiler).  Derived per kernel:
  mfma_pipe_cycles_per_simd   SQ_INSTS_MFMA x 64 cycles (v_mfma_f32_32x32x2_f32: 16 passes of 4 cycles) / 1024 SIMDs
  valu_per_mfma               non-MFMA VALU instructions per MFMA (SQ_INSTS_VALU counts MFMAs too: subtracted)
  issue_ceiling_pct           MFMA cycles / (MFMA cycles + co-issue cost of the VALU instructions), with the measured
                              per-class costs of profiles/r05_mfma_valu_hybrid.txt: transcendental 9.8 cycles, every
                              other VALU instruction 3.2 (low) .. 4.9 (high) -> [pessimistic, optimistic]
  mfma_util_pct               SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): what was measured
  wait_share_pct              SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: share of wave-resident time spent waiting on any counter
Kernels without MFMAs get `valu_busy_pct` instead: vector instructions x 4 cycles / (kernel cycles x 1024 SIMDs).
A kernel whose mfma_util is at its issue ceiling is issue-bound (fewer VALU instructions help); one far below it is
stall-bound (wait_share says how much of the waves' time is waiting)."""
import collections
import csv
import glob
import json
import sys

SIMDS = 1024
MFMA_CYCLES = 64
KEEP = ("attention_kernel", "stem_wino43_kernel", "conv3x3_wino_kernel", "gemm_nt_kernel", "gemm_rows512", "gemm_mlp",
        "det_head_kernel", "conv3x3", "nms_stream", "assign_", "select_kernel", "sample_kernel", "posenc_kernel")


def read(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main(dirs):
    acc = read(dirs)
    out = {}
    for k, c in sorted(acc.items()):
        if not any(s in k for s in KEEP):
            continue
        m = {name: sum(v) / len(v) for name, v in c.items()}
        mfma = m.get("SQ_INSTS_MFMA", 0.0)
        if mfma <= 0:
            # a kernel without matrix work (NMS, select, sampling, assignment tail): how busy is the VECTOR unit?  A wave64
            # instruction occupies a SIMD's 16 lanes for 4 cycles; kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
            gui = m.get("GRBM_GUI_ACTIVE", 0.0)
            if gui and m.get("SQ_INSTS_VALU"):
                out[k] = {"launches": len(c.get("SQ_INSTS_VALU", [])),
                          "insts_per_launch": {n.replace("SQ_INSTS_", "").lower(): round(v) for n, v in sorted(m.items())
                                               if n.startswith("SQ_INSTS_")},
                          "valu_busy_pct": round(100 * m["SQ_INSTS_VALU"] * 4 / (gui / 8 * SIMDS), 1),
                          "lds_per_valu": round(m.get("SQ_INSTS_LDS", 0.0) / m["SQ_INSTS_VALU"], 3),
                          "kernel_cycles": round(gui / 8)}
                if m.get("SQ_WAVE_CYCLES"):
                    out[k]["wait_share_pct"] = round(100 * m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 1)
            continue
        valu = max(m.get("SQ_INSTS_VALU", 0.0) - mfma, 0.0)
        trans = m.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
        mc = mfma * MFMA_CYCLES
        lo = mc / (mc + trans * 9.8 + (valu - trans) * 4.9)
        hi = mc / (mc + trans * 9.8 + (valu - trans) * 3.2)
        gui = m.get("GRBM_GUI_ACTIVE", 0.0)
        rec = {"launches": len(c.get("SQ_INSTS_MFMA", [])),
               "insts_per_launch": {n.replace("SQ_INSTS_", "").lower(): round(v) for n, v in sorted(m.items())
                                    if n.startswith("SQ_INSTS_")},
               "mfma_pipe_cycles_per_simd": round(mc / SIMDS),
               "valu_per_mfma": round(valu / mfma, 3), "trans_per_mfma": round(trans / mfma, 4),
               "lds_per_mfma": round(m.get("SQ_INSTS_LDS", 0.0) / mfma, 3),
               "salu_per_mfma": round(m.get("SQ_INSTS_SALU", 0.0) / mfma, 3),
               "issue_ceiling_pct": [round(100 * lo, 1), round(100 * hi, 1)],
               "mfma_util_pct": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * SIMDS) * 100, 1) if gui else None}
        if m.get("SQ_WAVE_CYCLES"):
            rec["wait_share_pct"] = round(100 * m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 1)
            rec["wait_lds_share_pct"] = round(100 * m.get("SQ_WAIT_INST_LDS", 0.0) / m["SQ_WAVE_CYCLES"], 1)
        for n in ("SQ_BUSY_CU_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES",
                  "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if n in m:
                rec.setdefault("raw_per_launch", {})[n] = round(m[n])
        out[k] = rec
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])

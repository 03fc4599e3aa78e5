"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE, one pass
each, as MI355X_MICROARCH.md prescribes) of `bench.py` into per-kernel, per-launch figures.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES \
        > profiles/r01_pmc_summary.json

gfx950 corrections applied: FETCH_SIZE is doubled (it reports exactly half the bytes of wide coalesced
reads; check: layernorm_gelu reads 134.2 MB algorithmic, raw counter 67.2 MB); WRITE_SIZE is exact.
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs).
eff_clock_ghz = GRBM_GUI_ACTIVE / 8 / dispatch duration of the same pass (the shader clock the kernel actually ran at under
the profiler: the chip lowers its clock under matrix load, so "fraction of the 2.4 GHz peak" and MfmaUtil differ)."""
import collections
import csv
import glob
import json
import sys


def agg(directory, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(directory + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return out


def durations(directory):
    out = collections.defaultdict(list)
    for f in glob.glob(directory + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in r and "End_Timestamp" in r:
                out[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(
                    (float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    return out


def main(fetch_dir, write_dir, sq_dir):
    dur = durations(sq_dir)
    f, w = agg(fetch_dir, "FETCH_SIZE"), agg(write_dir, "WRITE_SIZE")
    mb, ga = agg(sq_dir, "SQ_VALU_MFMA_BUSY_CYCLES"), agg(sq_dir, "GRBM_GUI_ACTIVE")
    res = {}
    for k in sorted(set(f) | set(w)):
        n = max(len(f.get(k, [])), 1)
        fetch = 2.0 * sum(f.get(k, [0])) / n * 1024
        write = sum(w.get(k, [0])) / max(len(w.get(k, [])), 1) * 1024
        busy = sum(mb.get(k, [0])) / max(len(mb.get(k, [])), 1)
        gui = sum(ga.get(k, [0])) / max(len(ga.get(k, [])), 1)
        res[k] = {"launches": len(f.get(k, [])), "fetch_bytes_per_launch": round(fetch),
                  "write_bytes_per_launch": round(write), "hbm_bytes_per_launch": round(fetch + write),
                  "mfma_util_pct": round(busy / (gui / 8 * 1024) * 100, 1) if gui else None}
        d = [(c, t) for c, t in dur.get(k, []) if t > 0]
        if d:
            res[k]["eff_clock_ghz"] = round(sum(c for c, _ in d) / 8 / sum(t for _, t in d), 3)
            res[k]["avg_ns_profiled"] = round(sum(t for _, t in d) / len(d))
    json.dump(res, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(*sys.argv[1:4])

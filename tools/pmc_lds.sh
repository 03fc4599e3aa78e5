# LDS bank-conflict and wait counters per kernel (diagnostic): one rocprofv3 --pmc pass per counter group
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $grp | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d gpurun_out/pmc_$tag -- python3 tools/bench_kernels.py --only ${1:-conv} > gpurun_out/pmc_$tag.log 2>&1 || true
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmc_SQ_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open("gpurun_out/pmc_lds_summary.txt", "w") as out:
    for k in sorted(agg):
        if not any(s in k for s in ("conv3x3", "gemm_nt", "attention_kernel")): continue
        a = {c: agg[k][c] / max(cnt[k][c], 1) for c in agg[k]}
        line = f"{k}: " + ", ".join(f"{c}={v:.3g}" for c, v in sorted(a.items()))
        if a.get("SQ_LDS_IDX_ACTIVE"): line += f" | bank_conflict_frac={a.get('SQ_LDS_BANK_CONFLICT',0)/a['SQ_LDS_IDX_ACTIVE']:.3f}"
        if a.get("SQ_WAVE_CYCLES"): line += f" | wait_lds/wave_cycles={a.get('SQ_WAIT_INST_LDS',0)/a['SQ_WAVE_CYCLES']:.3f}"
        print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/pmc_SQ_*

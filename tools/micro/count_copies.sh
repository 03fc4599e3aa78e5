# HIP API calls per bench step: difference of two traced runs with 1 and 3 steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 1 3; do
  rocprofv3 --hip-trace --output-format csv -d gpurun_out/hiptrace$n -- python3 bench.py --steps $n --warmup 0 --no-cpu-baseline --no-self-check --no-batch1 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, collections, glob
def count(n):
    f = glob.glob(f"gpurun_out/hiptrace{n}/*/*hip_api_trace.csv")[0]
    return collections.Counter(r["Function"] for r in csv.DictReader(open(f)))
a, b = count(1), count(3)
for k in sorted(b, key=lambda k: -(b[k] - a[k])):
    d = (b[k] - a[k]) / 2
    if d > 0: print(f"{d:8.1f} per step  {k}")
PY
rm -rf gpurun_out/hiptrace1 gpurun_out/hiptrace3

# per-shape durations of the GEMM launches inside the bench step (rocprofv3 kernel trace, bucketed by grid size)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/gemm_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gemm_trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-self-check --no-batch1 > gpurun_out/gemm_trace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/gemm_trace/*/*kernel_trace.csv")[0]
b = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "gemm_nt_kernel" in n or "rows512" in n or "attention_kernel" in n:
        key = (n.split("(")[0][-40:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        b[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(b.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k[0]:42s} grid {k[1]:>7s} {k[2]:>5s} {k[3]:>3s}  n {len(v):4d}  median {v2[len(v2)//2]:8.1f} us  total {sum(v)/1e3:8.2f} ms")
PY
rm -rf gpurun_out/gemm_trace

# sample shader clock and package power while bench.py runs (rocm-smi, 0.2 s period) -> gpurun_out/clock_probe.log
mkdir -p gpurun_out
( for i in $(seq 1 120); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/clock_probe.log 2>&1 &
SMI=$!
python bench.py --steps 200 --warmup 5 --no-self-check --no-cpu-baseline 2>/dev/null | tail -c 400
kill $SMI 2>/dev/null || true
wait $SMI 2>/dev/null || true
sort gpurun_out/clock_probe.log | uniq -c | sort -rn | head -12

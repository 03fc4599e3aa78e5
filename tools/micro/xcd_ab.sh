set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_primitives.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/xcd_tests.log 2>&1 || { tail -30 gpurun_out/xcd_tests.log; exit 1; }
tail -2 gpurun_out/xcd_tests.log
for v in 0 1 0 1; do
  echo "== GFC_XCD_REMAP=$v"
  GFC_XCD_REMAP=$v python tools/bench_kernels.py --only gemm 2>&1 | grep -v amdgpu.ids
  GFC_XCD_REMAP=$v python tools/bench_kernels.py --only wino 2>&1 | grep -v amdgpu.ids
  GFC_XCD_REMAP=$v python tools/bench_kernels.py --only attn 2>&1 | grep -v amdgpu.ids
done > gpurun_out/xcd_ab.log 2>&1
for v in 0 1 0 1; do GFC_XCD_REMAP=$v python bench.py --steps 6 --warmup 2 --no-self-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('remap=$v', d['value'], d['ms_per_step'])"; done

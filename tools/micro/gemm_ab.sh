set -e
python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "linear or batched or gemm" > gpurun_out/gemm_tests.log 2>&1 || { tail -30 gpurun_out/gemm_tests.log; exit 1; }
tail -1 gpurun_out/gemm_tests.log
for i in 1 2; do for l in prev ""; do echo "== lib ${l:-worktree}"; if [ -n "$l" ]; then export GFC_AMD_LIB=tools/ab_libs/libgfc_amd_$l.so; else unset GFC_AMD_LIB; fi; python tools/bench_kernels.py --only gemm 2>&1 | grep -v amdgpu.ids; done; done
unset GFC_AMD_LIB
bash tools/micro/lib_ab.sh prev 3

set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
tail -c 600 gpurun_out/r02_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof_stats -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-experimental --no-self-check > gpurun_out/r02_prof_stats.log 2>&1
echo stats done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/r02_pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-experimental --no-self-check > gpurun_out/r02_pmc_$c.log 2>&1
  echo pmc $c done
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/r02_pmc_SQ_VALU_MFMA_BUSY_CYCLES -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-experimental --no-self-check > gpurun_out/r02_pmc_SQ.log 2>&1
echo pmc sq done
python tools/pmc_summary.py gpurun_out/r02_pmc_FETCH_SIZE gpurun_out/r02_pmc_WRITE_SIZE gpurun_out/r02_pmc_SQ_VALU_MFMA_BUSY_CYCLES > gpurun_out/r02_pmc_summary.json
find gpurun_out/r02_prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_kernel_stats.csv
# keep the merged payload small
rm -rf gpurun_out/r02_pmc_FETCH_SIZE gpurun_out/r02_pmc_WRITE_SIZE gpurun_out/r02_pmc_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/r02_prof_stats
ls -la gpurun_out

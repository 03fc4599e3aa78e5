set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 600 gpurun_out/bench_final.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-experimental > gpurun_out/prof_stats.log 2>&1
echo stats done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-experimental > gpurun_out/pmc_$c.log 2>&1
  echo pmc $c done
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-experimental > gpurun_out/pmc_SQ.log 2>&1
echo pmc sq done
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES > gpurun_out/pmc_summary.json
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
# keep the merged payload small
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/prof_stats
ls -la gpurun_out

# Round-5 measurement set, one gpurun call:  bash tools/profile_r05.sh [tag]
#   bench line (default command) -> rocprofv3 kernel stats -> three separate PMC passes (FETCH_SIZE, WRITE_SIZE,
#   SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; never combined with tracing domains) -> per-kernel summary JSON.
# Copy gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc_summary.json} into profiles/ afterwards.
set -e
T=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 600 gpurun_out/${T}_bench.json
P="--no-cpu-baseline --no-self-check --no-batch1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof_stats -- python3 bench.py --steps 3 --warmup 3 $P > gpurun_out/${T}_prof_stats.log 2>&1
echo stats done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/${T}_pmc_$c -- python3 bench.py --steps 2 --warmup 1 $P > gpurun_out/${T}_pmc_$c.log 2>&1
  echo pmc $c done
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/${T}_pmc_SQ -- python3 bench.py --steps 2 --warmup 1 $P > gpurun_out/${T}_pmc_SQ.log 2>&1
echo pmc sq done
python tools/pmc_summary.py gpurun_out/${T}_pmc_FETCH_SIZE gpurun_out/${T}_pmc_WRITE_SIZE gpurun_out/${T}_pmc_SQ > gpurun_out/${T}_pmc_summary.json
find gpurun_out/${T}_prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats.csv
# keep the merged payload small
rm -rf gpurun_out/${T}_pmc_FETCH_SIZE gpurun_out/${T}_pmc_WRITE_SIZE gpurun_out/${T}_pmc_SQ gpurun_out/${T}_prof_stats
ls -la gpurun_out | tail -12

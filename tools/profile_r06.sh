# Round-6 measurement set, one gpurun call:  bash tools/profile_r06.sh [tag]
#   bench line (default command) -> rocprofv3 kernel stats -> separate PMC passes (never combined with tracing domains):
#   FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE (tools/pmc_summary.py), and the dynamic
#   instruction mix / wait cycles per kernel (tools/pmc_instmix.py: the executed counterpart of tools/issue_census.py).
# Copy gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc_summary.json,instmix.json} into profiles/ afterwards.
set -e
T=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 400 gpurun_out/${T}_bench.json
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
i=0
for group in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" \
             "SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_CVT" \
             "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS" \
             "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $group -d gpurun_out/${T}_mix_$i -- python3 bench.py --steps 2 --warmup 1 $P > gpurun_out/${T}_mix_$i.log 2>&1 || { echo "mix pass $i failed"; tail -5 gpurun_out/${T}_mix_$i.log; }
  echo mix $i done
done
python tools/pmc_instmix.py gpurun_out/${T}_pmc_SQ gpurun_out/${T}_mix_* > gpurun_out/${T}_instmix.json
find gpurun_out/${T}_prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats.csv
# keep the merged payload small
rm -rf gpurun_out/${T}_pmc_FETCH_SIZE gpurun_out/${T}_pmc_WRITE_SIZE gpurun_out/${T}_pmc_SQ gpurun_out/${T}_prof_stats gpurun_out/${T}_mix_[0-9]
ls -la gpurun_out | tail -12

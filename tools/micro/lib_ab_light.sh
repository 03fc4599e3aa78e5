# same-box A/B of two builds of the library on the timed region only:  bash tools/micro/lib_ab_light.sh <name of libgfc_amd_<name>.so> [rounds]
name=$1; rounds=${2:-2}
for i in $(seq $rounds); do
  for l in $name ""; do
    if [ -n "$l" ]; then export GFC_AMD_LIB=tools/ab_libs/libgfc_amd_$l.so; else unset GFC_AMD_LIB; fi
    python bench.py --steps 10 --warmup 3 --no-self-check --no-cpu-baseline --no-batch1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; k=r['kernels']
print('lib=${l:-worktree}', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step; attention', r['avg_launch_ms'], 'ms self', k[0]['avg_launch_ms'], 'cross', k[1]['avg_launch_ms'], 'stem', k[2]['avg_launch_ms'])"
  done
done

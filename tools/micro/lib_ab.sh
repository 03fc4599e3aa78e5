# same-box A/B of two builds of the library on the whole bench:  bash tools/micro/lib_ab.sh <name of libgfc_amd_<name>.so> [rounds]
name=$1; rounds=${2:-3}
for i in $(seq $rounds); do
  for l in $name ""; do
    if [ -n "$l" ]; then export GFC_AMD_LIB=tools/ab_libs/libgfc_amd_$l.so; else unset GFC_AMD_LIB; fi
    python bench.py --steps 6 --warmup 2 --no-self-check --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lib=${l:-worktree}', d['value'], d['ms_per_step'])"
  done
done

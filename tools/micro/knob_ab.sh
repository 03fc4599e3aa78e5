# same-box A/B of one environment knob on the whole bench:  bash tools/micro/knob_ab.sh GFC_GEMM_EPI 0 1 [rounds]
knob=$1; a=$2; b=$3; rounds=${4:-3}
for i in $(seq $rounds); do
  for v in $a $b; do
    env $knob=$v python bench.py --steps 6 --warmup 2 --no-self-check --no-cpu-baseline --no-batch1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$knob=$v', d['value'], d['ms_per_step'])"
  done
done

# A/B of one environment switch on ONE box: bash tools/ab_bench.sh VAR "val_a val_b" [bench.py flags]
VAR=$1; VALS=$2; shift 2
for round in 1 2; do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fp32-steps 0 --no-parity-pass --kernel-steps 0 "$@" 2>/dev/null \
      | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$VAR=$v', d['value'], d['ms_per_step'])"
  done
done

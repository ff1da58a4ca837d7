# A/B of one environment switch over tools/kernel_bench.py rows on ONE box: bash tools/ab_kernel.sh VAR "val_a val_b ..." "<--only filter>" [grep pattern]
VAR=$1; VALS=$2; ONLY=$3; PAT=${4:-ms}
for v in $VALS; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 300 python tools/kernel_bench.py --only "$ONLY" --iters 10 2>&1 | grep -v '^{' | grep "$PAT"
done

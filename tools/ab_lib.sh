# A/B of whole-library variants (compile-time switches) over tools/kernel_bench.py rows on ONE box:
#   bash tools/ab_lib.sh "<variant .so files>" "<--only filter>" [grep pattern]
# each variant is copied over the package's libawseg_hip.so for its run; the shipped library is restored at the end.
P=adverse_weather_semantic_segmentation_robustness_benchmark_amd
VARS=$1; ONLY=$2; PAT=${3:-ms}
cp $P/libawseg_hip.so /tmp/libawseg_hip.orig.so
trap 'cp /tmp/libawseg_hip.orig.so $P/libawseg_hip.so' EXIT      # an interrupted run must not leave an ablation build as the shipped library
echo "== shipped"
timeout -k 10 300 python tools/kernel_bench.py --only "$ONLY" --iters 10 2>&1 | grep -v '^{' | grep "$PAT"
for v in $VARS; do
  echo "== $v"
  cp $v $P/libawseg_hip.so
  timeout -k 10 300 python tools/kernel_bench.py --only "$ONLY" --iters 10 2>&1 | grep -v '^{' | grep "$PAT"
done
cp /tmp/libawseg_hip.orig.so $P/libawseg_hip.so

set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mixffn" -rP > gpurun_out/s7_t1.log 2>&1; echo "rc=$?" >> gpurun_out/s7_t1.log; grep -n "mixffn fused\|passed\|failed\|rc=\|Error" gpurun_out/s7_t1.log | tail -14
timeout -k 10 300 python tools/kernel_bench.py --iters 10 --only "mixffn" > gpurun_out/s7_kb.log 2>&1; grep -v '^{' gpurun_out/s7_kb.log | tail -3
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp32-steps 0 --no-parity-pass > gpurun_out/s7_bench.json 2> gpurun_out/s7_bench.err
python - <<'PY'
import json
for f in ("gpurun_out/s7_bench.json",):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d.get("resident_frames"))
        for k in d["kernels"][:12]: print("   ", k["kernel"], k["launches_per_step"], k["avg_ms"], k["frac"], k["time_share_of_step"])
    except Exception as e: print(f, "ERR", e)
PY

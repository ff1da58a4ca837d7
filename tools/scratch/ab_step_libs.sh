P=adverse_weather_semantic_segmentation_robustness_benchmark_amd
for r in 1 2; do for v in base few; do cp tmp_variants/lib_$v.so $P/libawseg_hip.so; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fp32-steps 0 --no-parity-pass --kernel-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['value'], d['ms_per_step'])"; done; done
cp tmp_variants/lib_few.so $P/libawseg_hip.so

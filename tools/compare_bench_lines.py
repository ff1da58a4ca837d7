#!/usr/bin/env python3
"""compare_bench_lines.py a.json b.json: the `miou` dicts of two bench.py lines (e.g. N=1 and N=2) must be identical,
key for key, bit for bit (SURVEY §8(d) parity gate: pooled mIoU from the fixed global sample set at any GPU count)."""
import json
import sys


def last_line(p):
    return json.loads([l for l in open(p).read().splitlines() if l.startswith("{")][-1])


a, b = last_line(sys.argv[1]), last_line(sys.argv[2])
bad = [k for k in sorted(set(a["miou"]) | set(b["miou"])) if a["miou"].get(k) != b["miou"].get(k)]
print(f"n_gpus {a['n_gpus']} vs {b['n_gpus']}: {len(a['miou'])} keys, {len(bad)} differ")
for k in bad:
    print(f"  {k}: {a['miou'].get(k)!r} vs {b['miou'].get(k)!r}")
raise SystemExit(1 if bad else 0)

#!/usr/bin/env python3
"""Post-process rocprofv3 CSV output (gpurun_out/...) into the summaries committed under profiles/.

  python tools/make_profiles.py kernel-table <trace_dir> <fetch_dir> <write_dir> <out.csv> "<header comment>"
      per-kernel calls / average duration from a --kernel-trace run plus HBM traffic per dispatch from two
      separate --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch, median).  hbm_read_MB applies the gfx950
      x2 correction for wide coalesced reads (MI355X_MICROARCH.md §HBM).
  python tools/make_profiles.py step <trace_dir> <marker_kernel_substring> <out.csv> "<header comment>"
      kernels of ONE bench step, cut out of the kernel trace between two consecutive launches of the marker kernel.
  python tools/make_profiles.py check-log <kernel_bench.log | bench line .json> [...]
      fail (exit 1) if any row of a kernel_bench / bench.py JSON line claims more than 100 % of its peak — such a row
      measured the wrong kernel or priced the wrong work (round 1 committed "181 % of the fp32 MFMA peak" this way).
"""
import collections
import csv
import glob
import statistics
import sys


def rows_of(d, pattern):
    out = []
    for f in glob.glob(f"{d}/**/*{pattern}*.csv", recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def short(name, n=160):
    return name.replace("void ", "", 1)[:n]


def kernel_table(trace_dir, fetch_dir, write_dir, out, header, stat="median"):
    dur = collections.defaultdict(list)
    for r in rows_of(trace_dir, "kernel_trace"):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    pmc = {}
    for tag, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        acc = collections.defaultdict(list)
        for r in rows_of(d, "counter_collection"):
            if r["Counter_Name"] == tag:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        pmc[tag] = {k: (statistics.median(v) if stat == "median" else sum(v) / len(v)) for k, v in acc.items()}
    with open(out, "w") as f:
        f.write(f"# {header}\n")
        f.write("# traffic: separate passes rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (KB per dispatch, " + stat + "); hbm_read_MB applies "
                "the gfx950 x2 correction for wide coalesced reads (MI355X_MICROARCH.md §HBM)\n")
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "avg_us", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_read_MB_corrected", "hbm_write_MB"])
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            if "anonymous namespace" not in k or "at::native" in k:
                continue
            fk, wk = pmc["FETCH_SIZE"].get(k), pmc["WRITE_SIZE"].get(k)
            w.writerow([short(k), len(v), round(sum(v) / len(v), 2), "" if fk is None else fk, "" if wk is None else wk,
                        "" if fk is None else round(2 * fk / 1024, 2), "" if wk is None else round(wk / 1024, 2)])


def step(trace_dir, marker, out, header):
    rows = rows_of(trace_dir, "kernel_trace")
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    # the marker runs once per step: take the window between the last two of them (steady state)
    a, b = marks[-2], marks[-1]
    win = rows[a:b]
    t0, t1 = int(win[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    agg = collections.defaultdict(list)
    for r in win:
        agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out, "w") as f:
        f.write(f"# {header}\n# one timed step: window {(t1 - t0) / 1e6:.3f} ms, {len(win)} launches, kernel time {sum(sum(v) for v in agg.values()) / 1e3:.3f} ms\n")
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([short(k), len(v), round(sum(v) / 1e3, 4), round(sum(v) / len(v), 2)])


def check_log(paths):
    import json
    bad = []
    for p in paths:
        for line in open(p):
            line = line.strip()
            if not line.startswith("{"):
                continue
            try:
                d = json.loads(line)
            except ValueError:
                continue
            for r in d.get("rows", []) + d.get("kernels", []) + ([d["roofline"]] if d.get("roofline") else []):
                frac = r.get("frac_of_peak", r.get("frac"))
                if frac is not None and frac > 1.0:
                    bad.append(f"{p}: {r.get('kernel')} frac {frac}")
    if bad:
        raise SystemExit("rows above their peak:\n  " + "\n  ".join(bad))
    print(f"check-log: {len(paths)} file(s), no row above its peak")


if __name__ == "__main__":
    if sys.argv[1] == "check-log":
        check_log(sys.argv[2:])
    elif sys.argv[1] == "kernel-table":
        kernel_table(*sys.argv[2:8])
    elif sys.argv[1] == "step":
        step(*sys.argv[2:6])
    else:
        raise SystemExit(__doc__)

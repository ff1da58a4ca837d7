"""Ordered launch list of the last whole step in a rocprofv3 kernel trace of bench.py (steps are delimited by weather_batch_kernel).
python tools/scratch/step_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "weather_batch_kernel" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n)[:70]
prev_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
print("launches", len(step), "window us", (prev_end - t0) / 1e3)

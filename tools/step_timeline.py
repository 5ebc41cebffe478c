"""Launch-by-launch timeline of one training step out of a rocprofv3 kernel trace:
python tools/step_timeline.py <kernel_trace.csv> [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "grid_build" in r["Kernel_Name"]]
s, e = idx[-3], idx[-2]
t0 = int(rows[s]["Start_Timestamp"])
for r in rows[s:e]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("cutie::", "")[:64]
    if d >= min_us:
        print("%8.1f us %7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, n))
print("step span us", (int(rows[e]["Start_Timestamp"]) - t0) / 1e3)

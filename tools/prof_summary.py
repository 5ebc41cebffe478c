"""Per-step summary of a rocprofv3 --kernel-trace --stats kernel_stats.csv: python tools/prof_summary.py <csv> <steps>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 33
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step: %.3f ms" % (tot / steps / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-92s %5.1f/step %8.1f us %5.1f%%" % (r["Name"][:92], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                  float(r["TotalDurationNs"]) / tot * 100))

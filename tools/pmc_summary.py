"""Per-kernel averages of a rocprofv3 --pmc run: python tools/pmc_summary.py <counter_collection.csv> [...] > out.json
Several CSVs (separate passes) are merged by kernel name."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|cutie::|void ", "", name)
    return name.split("(")[0].strip()


acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Counter_Name"]].add((path, r["Dispatch_Id"]))
out = {}
for k in acc:
    out[k] = {c: {"launches": len(disp[k][c]), "avg": acc[k][c] / max(1, len(disp[k][c]))} for c in acc[k]}
json.dump(out, sys.stdout, indent=1, sort_keys=True)

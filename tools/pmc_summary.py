"""Per-kernel means of rocprofv3 --pmc counters.  usage: python tools/pmc_summary.py <dir> [name-filter]"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "ep::"
acc = defaultdict(lambda: defaultdict(list))
for f in root.rglob("*counter_collection.csv"):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if flt not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace("ep::", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    parts = [f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(acc[k].items())]
    n = len(next(iter(acc[k].values())))
    print(f"{k:40s} n={n:4d}  " + "  ".join(parts))

"""Per-kernel averages of the SQ counters collected by tools/pmc_sq.sh (values per wave where that makes sense)."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(out, "g*.csv"))):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "esdg::" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    waves = avg.get("SQ_WAVES", 0) or 1
    print(k, f"waves={waves:.0f}")
    for c in sorted(avg):
        if c != "SQ_WAVES":
            print(f"   {c:24s} total {avg[c]:14.0f}   per wave {avg[c] / waves:10.1f}")

"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel-name prefix, sum each counter over dispatches."""
import collections
import csv
import sys

for path in sys.argv[2:]:
    acc = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if sys.argv[1] in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(f"{k:34s} dispatches={n[k]:5d} sum={acc[k]:.6g}")

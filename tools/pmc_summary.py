"""Summarise rocprofv3 --pmc counter_collection.csv files.

  pmc_summary.py <kernel-name-substring> file.csv ...   sum of each counter over the matching dispatches
  pmc_summary.py per-kernel file.csv                    per kernel name: dispatches, sum and mean of each counter
"""
import collections
import csv
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()[:80]


if sys.argv[1] == "per-kernel":
    acc = collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(sys.argv[2])):
        k = (short(r.get("Kernel_Name", "")), r["Counter_Name"])
        acc[k] += float(r["Counter_Value"])
        n[k] += 1
    for (kn, c) in sorted(acc):
        print(f"{kn:80s} {c:22s} dispatches={n[(kn, c)]:6d} sum={acc[(kn, c)]:.6g} mean={acc[(kn, c)] / n[(kn, c)]:.6g}")
else:
    for path in sys.argv[2:]:
        acc = collections.defaultdict(float)
        n = collections.Counter()
        for r in csv.DictReader(open(path)):
            if sys.argv[1] in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
        for k in sorted(acc):
            print(f"{k:34s} dispatches={n[k]:5d} sum={acc[k]:.6g}")

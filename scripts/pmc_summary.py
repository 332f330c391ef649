#!/usr/bin/env python3
"""Mean counter value per kernel (largest grid only) from a rocprofv3 --pmc output directory.  usage: pmc_summary.py DIR [name-regex]"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if pat and not pat.search(name):
            continue
        key = (re.sub(r"\(.*", "", name)[:90], int(row["Grid_Size"]))
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key in sorted(acc):
    vals = acc[key]
    print(key[0], "grid", key[1], "n", len(next(iter(vals.values()))))
    for c in sorted(vals):
        print(f"    {c:32s} {sum(vals[c]) / len(vals[c]):16.1f}")

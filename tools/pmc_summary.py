#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel name over all passes found under a directory."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "prefill" not in name and "decode" not in name:
            continue
        short = name.split("(")[0].split("::")[-1][:60]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {sum(v)/len(v):18.1f}   (n={len(v)})")

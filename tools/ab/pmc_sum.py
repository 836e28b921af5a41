#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs under a directory (argv[1]); optional argv[2] = kernel-name filter."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, "n=%d" % len(next(iter(cs.values()))))

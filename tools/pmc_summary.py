#!/usr/bin/env python
"""Sum rocprofv3 --pmc counter CSVs per kernel name: python tools/pmc_summary.py <dir> [name-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if flt and flt not in name: continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); calls[name].add((f, r["Dispatch_Id"]))
for name, cs in acc.items():
    print(name, "dispatches", len(calls[name]))
    for c, v in sorted(cs.items()): print("   %-36s %.6g" % (c, v))

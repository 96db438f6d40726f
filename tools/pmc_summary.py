#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files (development aid): python tools/pmc_summary.py file.csv [filter]"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0][-48:]
    if flt and flt not in name:
        continue
    rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key); calls[name] += 1
for name, c in rows.items():
    print(name, "dispatches", calls[name])
    for k, v in sorted(c.items()):
        print("   %-32s %16.0f  per dispatch %14.1f" % (k, v, v / max(calls[name], 1)))

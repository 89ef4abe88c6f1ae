#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run: per kernel name, dispatch count and the mean counter value per dispatch.
usage: pmc_summary.py DIR COUNTER"""
import csv
import glob
import os
import sys

d, counter = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = {}
for fn in files:
    with open(fn, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row.get("Kernel_Name", "?").split("(")[0]
            if not (name.startswith("k_") or name.startswith("void k_")):
                continue
            a = acc.setdefault(name, {})
            did = row.get("Dispatch_Id")
            a[did] = a.get(did, 0.0) + float(row["Counter_Value"])
print("kernel,dispatches,%s_mean_per_dispatch,%s_total" % (counter, counter))
for name, a in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
    tot = sum(a.values())
    print("%s,%d,%.1f,%.1f" % (name.replace(",", ";"), len(a), tot / max(1, len(a)), tot))

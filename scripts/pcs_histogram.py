#!/usr/bin/env python3
"""rocprofv3 PC sampling output (a directory) -> histogram of the sampled instructions of the kernel whose name contains KERNEL.
usage: pcs_histogram.py DIR KERNEL"""
import collections
import csv
import glob
import os
import sys

out, kern = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(out, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
print("pc sampling files:", files)
hist = collections.Counter(); total = 0; cols = None
# dispatch id -> kernel name from the kernel trace, when the sampling rows do not carry the name
names = {}
for fn in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(fn, newline="")):
        names[row.get("Dispatch_Id")] = row.get("Kernel_Name", "")
for fn in files:
    with open(fn, newline="") as f:
        rd = csv.DictReader(f)
        cols = rd.fieldnames
        for row in rd:
            total += 1
            name = row.get("Kernel_Name") or row.get("kernel_name") or names.get(row.get("Dispatch_Id") or row.get("dispatch_id"), "")
            if name and kern not in name:
                continue
            key = row.get("Instruction") or row.get("instruction") or row.get("Instruction_Comment")
            if not key:
                key = "code object %s + %s" % (row.get("Code_Object_Id", "?"), row.get("Code_Object_Offset", row.get("PC", "?")))
            hist[key] += 1
print("samples:", total, "columns:", cols)
n = sum(hist.values())
print("samples in kernels matching %r: %d" % (kern, n))
for key, c in hist.most_common(120):
    print("%6d %5.1f%%  %s" % (c, 100.0 * c / max(1, n), key[:160]))

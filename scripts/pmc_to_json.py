#!/usr/bin/env python3
"""profiles/<tag>_pmc_FETCH_SIZE.csv + _WRITE_SIZE.csv (+ optional SQ_INSTS_VALU) -> profiles/pmc_traffic.json, the file bench.py
reads for roofline.traffic.  HBM bytes per launch = 2 x FETCH_SIZE KiB (gfx950 counts 128-byte read requests as 64 bytes,
MI355X_MICROARCH.md section HBM) + WRITE_SIZE KiB.   usage: pmc_to_json.py TAG"""
import csv
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES"):
    fn = os.path.join(root, "profiles", "%s_pmc_%s.csv" % (tag, ctr))
    if not os.path.exists(fn):
        continue
    for row in csv.DictReader(open(fn)):
        name = row["kernel"].replace("void ", "").split("<")[0]
        if "; true>" in row["kernel"]:                      # k_l2_sweep<true> is the wide pass
            name += "_hard"
        if row["kernel"].endswith("<true>"):
            name += "_wide"
        out.setdefault(name, {})[ctr] = float(row["%s_mean_per_dispatch" % ctr])
for name, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
out["_note"] = "per launch at bench.py's default workload (1 M reads x 10 kbp per GPU); FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them; read side doubled per the gfx950 correction; source tag " + tag
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:1500])

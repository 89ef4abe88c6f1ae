#!/usr/bin/env python3
"""profiles/<tag>_pmc_FETCH_SIZE.csv + _WRITE_SIZE.csv (+ optional SQ_INSTS_VALU) -> profiles/pmc_traffic.json[WORKLOAD], the file
bench.py reads for roofline.traffic / roofline.valu of that workload (bench.py names it as `traffic_source`: the counters come from
committed rocprofv3 --pmc passes, not from the bench run).  HBM bytes per launch = 2 x FETCH_SIZE KiB (gfx950 counts 128-byte read
requests as 64 bytes, MI355X_MICROARCH.md section HBM) + WRITE_SIZE KiB.
usage: pmc_to_json.py TAG WORKLOAD     (WORKLOAD: configs1 | configs2 | configs3 | configs4 | northstar | northstar_seg10000 | northstar_repeat_rich)"""
import csv
import json
import os
import sys

tag, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES"):
    fn = os.path.join(root, "profiles", "%s_pmc_%s.csv" % (tag, ctr))
    if not os.path.exists(fn):
        continue
    for row in csv.DictReader(open(fn)):
        full = row["kernel"].replace("void ", "")
        name = full.split("<")[0]
        targs = full[len(name) + 1:].rstrip(">").replace(";", ",").split(",") if "<" in full else []
        if name == "k_l2_sweep" and targs and targs[0].strip() == "true":      # k_l2_sweep<true, JB, LPW> is the wide-cell pass
            name += "_wide"
        d = out.setdefault(name, {})
        # several instantiations of one template (k_lookup_l1<128> / <512>): keep the one with more dispatches' worth of traffic
        v = float(row["%s_mean_per_dispatch" % ctr])
        d[ctr] = max(v, d.get(ctr, 0.0))
for name, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
out["_source"] = tag
sys.path.insert(0, root)
import bench  # noqa: E402
out["_csrc_sha16"] = bench.csrc_sha16()                      # the kernel sources these counters belong to (bench.py compares with the tree it runs on)
out["_collected"] = os.environ.get("MM_PMC_COLLECTED", "round 6")
out["_note"] = ("per launch at bench.py's `%s` workload; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them; read side doubled per the gfx950 "
                "correction; from profiles/%s_pmc_*.csv" % (workload, tag))
path = os.path.join(root, "profiles", "pmc_traffic.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
if "k_sketch_fast" in allw:                                  # the round-2 layout (one workload, kernels at top level)
    allw = {"configs1": allw}
allw[workload] = out
json.dump(allw, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: out[k] for k in sorted(out) if k in ("k_sketch_fast", "k_lookup_l1", "k_l2_locate", "k_l2_sweep", "_source")}, indent=1))

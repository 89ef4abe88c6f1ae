#!/usr/bin/env python3
"""the gist of a bench.py full record (or an old one-line record) on a few terminal lines (gpurun only returns the tail of the output).  usage: bench_digest.py FILE"""
import json
import sys


def kern(k):
    return "  ".join("%s %.2f" % (n, v["ms_per_step"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"]))


def one(tag, d):
    if not isinstance(d, dict) or "value" not in d:
        print("%s: %s" % (tag, json.dumps(d)[:300])); return
    r = d.get("roofline") or {}
    print("%s: %.1f %s, %.2f ms/step, passes %s" % (tag, d["value"], d.get("unit", ""), d.get("ms_per_step", 0), {k: v for k, v in (d.get("passes") or {}).items() if k != "note"}))
    if d.get("kernels"):
        print("   kernels (ms/step): " + kern(d["kernels"]))
    if r:
        i = r.get("int") or {}
        print("   roofline: bound %s frac %s (mix %s) | hbm frac %s | hash-only %s ms, sketch_kernel_frac %s" %
              (r.get("bound"), r.get("frac"), r.get("frac_vs_measured_mix"), (r.get("hbm") or {}).get("frac"), i.get("hash_only_ms"), i.get("sketch_kernel_frac")))
    for k in ("fragments", "interval_points_per_fragment", "l1_candidates_per_fragment", "hard_list_share", "hbm_point_path_share", "index_build_s"):
        if k in d:
            print("   %s: %s" % (k, d[k]))


raw = open(sys.argv[1]).read()
try:
    d = json.loads(raw)                                    # the full record (profiles/bench_last_full.json), indented
except ValueError:
    txt = [l for l in raw.splitlines() if l.startswith("{")]
    if not txt:
        print("no JSON line in", sys.argv[1]); sys.exit(0)
    d = json.loads(txt[-1])
print("workload:", d["config"]["workload"])
one("headline", d)
print("   config:", {k: d["config"].get(k) for k in ("fragments_per_gpu", "resident_batches", "l1_candidates_per_gpu", "candidate_mappings_per_gpu", "hard_list_fragments", "index_build_s", "host_synchronisations_last_pass")})
if "host_path" in d:
    print("host_path:", {k: d["host_path"].get(k) for k in ("device_ms", "download_ms", "host_ms", "host_threads", "gbps_pipelined", "error")})
if "cpu_baseline" in d:
    print("cpu_baseline:", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "threads", "kind", "error")})
if "e2e" in d:
    e = d["e2e"]
    print("e2e:", {k: e.get(k) for k in ("value", "map_s", "index_s", "paf_lines", "error")}, e.get("stages"), {k: v for k, v in (e.get("device_stage") or {}).items() if k != "note"})
if isinstance(d.get("configs2"), dict):
    one("configs2 (resident)", d["configs2"])
    e = d["configs2"].get("e2e") or {}
    print("   configs2 FASTA -> PAF:", {k: e.get(k) for k in ("value", "map_s", "paf_lines", "error")}, e.get("stages"))
ns = d.get("north_star_target")
if ns:
    one("north_star segLength 5000", ns)
    print("   cpu_baseline:", {k: (ns.get("cpu_baseline") or {}).get(k) for k in ("value", "cores", "threads", "index_build_s", "error")})
    one("north_star segLength 10000", ns.get("segLength_10000"))
    one("north_star repeat_rich", ns.get("repeat_rich"))

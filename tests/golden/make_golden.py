#!/usr/bin/env python3
"""Generates tests/golden/*.json|npz by running the REAL reference (oracle/_ref/libmashmap_ref.so, compiled from the
sources under /root/reference by oracle/Makefile) on the seeded inputs of cases.py.  Run in the build container:

    python tests/golden/make_golden.py

The fixtures are what pins the oracle (tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_golden.py) on
machines where /root/reference does not exist."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import mmutil as U          # noqa: E402
import cases as CS          # noqa: E402


def main():
    U.build_oracle()
    assert U.Ref.available(), "oracle/_ref not built: /root/reference missing?"
    ref = U.Ref()
    out = {}
    out["hashes"] = [[s.decode(), str(ref.get_hash(s))] for s in CS.hash_inputs()]
    out["sketch"] = {}
    for name, k, s, seq in CS.sketch_cases():
        out["sketch"][name] = [[str(h), a, b, c, d] for (h, a, b, c, d) in ref.sketch_sequence(seq, k, s, 7)]
    mm = {}
    for name, k, w, s, seq in CS.minmer_cases():
        mm[name] = ref.add_minmers(seq, k, w, s, 3)
    np.savez_compressed(os.path.join(HERE, "minmers.npz"), **mm)
    out["stats"] = {
        "j2md_130": [repr(float(ref.f("j2md")(i / 130.0, 19))) for i in range(0, 131)],
        "md2j": [repr(float(ref.f("md2j")(d / 100.0, 19))) for d in range(0, 40)],
        "md_lower_bound": [repr(float(ref.f("md_lower_bound")(d / 100.0, s, 19, 0.95))) for d in (1, 5, 10, 15, 20) for s in (20, 130, 498)],
        "min_hits_relaxed": {"%d_%d" % (s, int(pi * 100)): [int(ref.f("min_hits_relaxed")(q, 19, pi)) for q in range(1, s + 1)]
                             for s, pi in ((130, 0.85), (498, 0.80), (40, 0.95), (310, 0.85))},
        "recommended_sketch_size": {"%d_%d_%d" % (int(pi * 100), L, rs): int(ref.f("recommended_sketch_size")(19, pi, L, rs))
                                    for pi, L, rs in ((0.85, 5000, 100000000), (0.95, 10000, 3000000000), (0.85, 5000, 3000000000),
                                                      (0.90, 5000, 12000000), (0.85, 5000, 18446744072414584320))},
    }
    contigs, reads, P = CS.session_case()
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "ref.fa")
        U.write_fasta(fa, contigs)
        h = ref.session([fa], P["k"], P["segLength"], P["sketchSize"], P["pi"], U.FILTER_MAP, U.FLAG_HG, b"\0", P["kmerPct"])
        idx = ref.index_array(h)
        keys, counts = ref.keys(h)
        sess = {"n_minmers": int(len(idx)), "n_keys": int(len(keys)), "freq_threshold": int(ref.f("session_freq_threshold")(h)),
                "cutoffs": ref.cutoffs(h), "fragments": []}
        np.savez_compressed(os.path.join(HERE, "session_index.npz"), minmers=idx, keys=keys, counts=counts)
        for ri, (name, a) in enumerate(reads):
            for off, ln in CS.fragments_of(len(a), P["segLength"]):
                e = ref.map_fragment(h, a[off:off + ln], ri, name.encode(), len(a), P["sketchSize"])
                sess["fragments"].append({
                    "read": ri, "off": off, "len": ln, "sketchSize": e["sketchSize"], "rawSketchSize": e["rawSketchSize"],
                    "minimumHits": e["minimumHits"], "kmerComplexity": repr(e["kmerComplexity"]),
                    "sketch_head": [[str(x[0]), x[4]] for x in e["sketch"][:6]], "sketch_last": str(e["sketch"][-1][0]) if e["sketch"] else "",
                    "points": [list(p[:3]) for p in e["points"]], "l1": [list(x) for x in e["l1"]], "l2": [list(x) for x in e["l2"]],
                    "maps_i": [list(x) for x in e["maps_i"]], "maps_f": [[repr(float(v)) for v in x] for x in e["maps_f"]]})
        ref.free(h)
    out["session"] = sess
    # end-to-end PAF fixtures: the reference binary's own output for the command lines of cases.paf_cases()
    import subprocess
    pafdir = os.path.join(HERE, "paf")
    os.makedirs(pafdir, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        for name, refrec, qrec, extra in CS.paf_cases():
            rf = os.path.join(td, name + ".ref.fa"); U.write_fasta(rf, refrec)
            args = [U.REF_BIN, "-r", rf, "-o", os.path.join(pafdir, name + ".paf"), "-t", "2"] + extra
            if qrec is not None:
                qf = os.path.join(td, name + ".q.fa"); U.write_fasta(qf, qrec); args += ["-q", qf]
            subprocess.run(args, check=True, capture_output=True)
    with tempfile.TemporaryDirectory() as td:
        for name, ref_files, q_files, extra in CS.paf_list_cases():
            rl, ql = os.path.join(td, name + ".rl"), os.path.join(td, name + ".ql")
            with open(rl, "w") as f:
                for i, recs in enumerate(ref_files):
                    fn = os.path.join(td, "%s.ref%d.fa" % (name, i)); U.write_fasta(fn, recs); f.write(fn + "\n")
            with open(ql, "w") as f:
                for i, recs in enumerate(q_files):
                    fn = os.path.join(td, "%s.q%d.fa" % (name, i)); U.write_fasta(fn, recs); f.write(fn + "\n")
            subprocess.run([U.REF_BIN, "--rl", rl, "--ql", ql, "-o", os.path.join(pafdir, name + ".paf"), "-t", "2"] + extra, check=True, capture_output=True)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("golden: %d hashes, %d sketches, %d minmer cases, %d fragments" %
          (len(out["hashes"]), len(out["sketch"]), len(mm), len(sess["fragments"])))


if __name__ == "__main__":
    main()

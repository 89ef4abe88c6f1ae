#!/bin/bash
# device stage of the product's FASTA -> PAF path per 512 Mbp batch: steady-state passes (default) against MM_NO_STEADY=1 (a sizing pass per
# batch, six host read-backs).  usage: scripts/gpu_e2e_steady.sh TAG [reads]
TAG=${1:-e2e}; READS=${2:-500000}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 280 python scripts/e2e_fasta_paf.py --reads $READS > $OUT/steady.json 2> $OUT/steady.err
MM_NO_STEADY=1 timeout 120 python scripts/e2e_fasta_paf.py --reads $READS --reuse > $OUT/nosteady.json 2> $OUT/nosteady.err
python - $OUT <<'PY' | tee $OUT/log.txt
import re, sys, json, statistics as st
out = sys.argv[1]
for name in ("steady", "nosteady"):
    txt = open("%s/%s.err" % (out, name)).read()
    rows = [(int(n), float(t), float(k)) for n, t, k in re.findall(r"download of (\d+) candidate mappings\): ([0-9.e+-]+) s \(upload [0-9.e+-]+, kernels ([0-9.e+-]+)", txt)]
    full = [r for r in rows if r[0] > 0.9 * max(x[0] for x in rows)] if rows else []
    try: j = json.load(open("%s/%s.json" % (out, name)))
    except Exception: j = {}
    if full:
        print("%-9s %3d full batches: kernels median %.3f ms, mean %.3f ms, min %.3f ms; device stage median %.3f ms | FASTA->PAF %s Gbp/s, map_s %s"
              % (name, len(full), 1e3 * st.median(r[2] for r in full), 1e3 * st.mean(r[2] for r in full), 1e3 * min(r[2] for r in full), 1e3 * st.median(r[1] for r in full),
                 j.get("gbps_fasta_to_paf"), j.get("map_s")))
    else:
        print(name, "no device-stage lines", txt[-500:])
PY

#!/bin/bash
# where the wave cycles of the map kernels go: one rocprofv3 --pmc pass of SQ counters over one pass of the default workload
# (quad-cycle units; WAIT_ANY = parked on s_waitcnt, WAIT_INST_ANY = issue stall, ACTIVE_INST_* = issuing).  usage: scripts/gpu_sq_profile.sh TAG
TAG=${1:-sq}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
CS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
timeout 900 rocprofv3 --pmc $CS --kernel-trace --output-format csv -d $OUT/pmc -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-north-star > /dev/null 2> $OUT/pmc.err
for C in $CS; do python scripts/pmc_summary.py $OUT/pmc $C > $OUT/$C.csv; done
python - $OUT $CS <<'PY' | tee $OUT/log.txt
import csv, sys
out, cs = sys.argv[1], sys.argv[2:]
tab = {}
for c in cs:
    for row in csv.DictReader(open("%s/%s.csv" % (out, c))):
        tab.setdefault(row["kernel"], {})[c] = float(row["%s_mean_per_dispatch" % c])
print("%-44s " % "kernel" + " ".join("%14s" % c.replace("SQ_", "") for c in cs))
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    print("%-44s " % k[:44] + " ".join("%14.4g" % v.get(c, float("nan")) for c in cs))
PY
rm -rf $OUT/pmc

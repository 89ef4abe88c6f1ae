#!/bin/bash
# same-box hunt for a regression: one workload (scaled) through several trees / switches.  usage: scripts/gpu_regress.sh TAG WORKLOAD READS
TAG=${1:-rg}; WL=${2:-configs4}; READS=${3:-150000}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-40s %7.2f Gbp/s %8.3f ms/step | " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()) + " | index %.1f s" % d["config"]["index_build_s"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-path --workload $WL --reads $READS"
for T in r3_tree r11c_tree; do
  rm -rf /tmp/$T && mkdir -p /tmp/$T && tar xzf gpurun_aux/$T.tgz -C /tmp/$T
  ( cd /tmp/$T && timeout 900 $B > $OUT/b_$T.json 2> $OUT/b_$T.err ); summ "$WL $T" $OUT/b_$T.json | tee -a $OUT/log.txt
done
i=0
for V in "X=1" "MM_NO_STEADY=1" "MM_INDEX_NO_EARLY_UPLOAD=1"; do
  i=$((i+1)); env $V timeout 900 $B > $OUT/b_new$i.json 2> $OUT/b_new$i.err; summ "$WL new [$V]" $OUT/b_new$i.json | tee -a $OUT/log.txt
done

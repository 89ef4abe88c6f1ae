#!/bin/bash
# same-box A/B of an older tree (gpurun_aux/<name>.tgz: sources + built library) against this one.  usage: scripts/gpu_old_new.sh TAG NAME [workloads...]
TAG=${1:-on}; NAME=${2:-r11c_tree}; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf /tmp/oldtree && mkdir -p /tmp/oldtree && tar xzf gpurun_aux/$NAME.tgz -C /tmp/oldtree
summ() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-24s %7.2f Gbp/s %8.3f ms/step | " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()))
PY
}
for WL in "$@"; do
  for ROUND in 1 2; do
    for WHICH in old new; do
      D=$PWD; [ $WHICH = old ] && D=/tmp/oldtree
      ( cd $D && timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-north-star --workload $WL > $OUT/b_${WHICH}.json 2> $OUT/b_${WHICH}.err )
      summ "$WL $WHICH #$ROUND" $OUT/b_${WHICH}.json 2>&1 | tee -a $OUT/log.txt
    done
  done
done

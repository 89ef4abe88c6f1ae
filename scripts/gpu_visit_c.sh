#!/bin/bash
# visit: parity subset, index build phases (MM_DEBUG) with / without the early upload, the other BASELINE shapes at full size
TAG=${1:-vc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
scripts/gpu_ab.sh $TAG "" subset
for WL in northstar configs4; do
  for V in "X=1" "MM_INDEX_NO_EARLY_UPLOAD=1"; do
    echo "== index build $WL [$V]" | tee -a $OUT/log.txt
    env $V MM_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --workload $WL --reads 2000 2> $OUT/ix.err > /dev/null
    grep "\[mm\] index\|index build" $OUT/ix.err | tee -a $OUT/log.txt
  done
done
summ() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-12s %7.2f Gbp/s %8.3f ms/step | " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()) + " | index %.1f s syncs %s" % (d["config"]["index_build_s"], d["config"].get("host_synchronisations_per_pass")))
PY
}
for WL in configs3 configs4; do
  timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-path --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  summ $WL $OUT/bench_$WL.json 2>&1 | tee -a $OUT/log.txt
done

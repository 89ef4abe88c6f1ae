#!/bin/bash
# quick A/B of env switches on the default workload (no north star).  usage: scripts/gpu_quick.sh TAG "ENV1=.." "ENV2=.." ...
TAG=${1:-q}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for V in "X=1" "$@"; do
  env $V timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-north-star > $OUT/b.json 2> $OUT/b.err
  python - "$V" $OUT/b.json <<'PY' | tee -a $OUT/log.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("%-28s %7.2f Gbp/s %8.3f ms/step | " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()) + " | index %.2f s" % d["config"]["index_build_s"])
PY
done

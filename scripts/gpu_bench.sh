#!/bin/bash
# quick GPU visit: one bench line (no CPU baseline).  usage: scripts/gpu_bench.sh TAG [bench args...]
TAG=${1:-b}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
MM_DEBUG=1 timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
grep -v "^\[mm\] sketch" $OUT/bench.err | tail -8
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], "Gbp/s", d["ms_per_step"], "ms/step")
for k,v in d["kernels"].items(): print("  %-12s %8.2f ms" % (k, v["ms_per_step"]))
PY

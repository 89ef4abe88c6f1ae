#!/bin/bash
# One GPU-box visit for an A/B of an env switch: parity suite, a parity subset with the switch on, bench lines both ways.
# usage: scripts/gpu_ab.sh TAG "ENV=VAL ..." suite|subset|none [workloads...]   (workload "default" = the default bench line incl. north_star)
TAG=${1:-ab}; SW=${2:-}; SUITE=${3:-suite}; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
SUBSET="tests/test_gpu_map.py tests/test_gpu_golden.py tests/test_gpu_paf.py tests/test_gpu_multigpu.py tests/test_gpu_fullsize.py"
: > $OUT/log.txt
if [ "$SUITE" = suite ]; then
  echo "== pytest -m gpu" | tee -a $OUT/log.txt
  timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25 | tee -a $OUT/log.txt
elif [ "$SUITE" = subset ]; then
  echo "== parity subset" | tee -a $OUT/log.txt
  timeout 1500 python -m pytest $SUBSET -m gpu -x -q 2>&1 | grep -v "^$" | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 | tee -a $OUT/log.txt
fi
if [ -n "$SW" ]; then
  echo "== parity subset with $SW" | tee -a $OUT/log.txt
  env $SW timeout 1500 python -m pytest $SUBSET -m gpu -x -q 2>&1 | grep -v "^$" | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 | tee -a $OUT/log.txt
fi
summ() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
def line(tag, x):
    print("%-22s %7.2f Gbp/s %8.3f ms/step | " % (tag, x["value"], x["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in x["kernels"].items()))
line(d["config"]["workload"][:22], d)
ns = d.get("north_star_target")
if ns and "kernels" in ns:
    line("north_star", ns)
    if "segLength_10000" in ns and "kernels" in ns["segLength_10000"]: line("north_star seg10000", ns["segLength_10000"])
    print("index_build_s", ns.get("index_build_s"))
PY
}
for WL in "$@"; do
  for MODE in off on; do
    [ "$MODE" = on ] && [ -z "$SW" ] && continue
    E=""; [ "$MODE" = on ] && E="$SW"
    F=$OUT/bench_${WL}_$MODE.json
    if [ "$WL" = default ]; then env $E timeout 1200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $F 2> $OUT/bench_${WL}_$MODE.err
    else env $E timeout 1200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-path --workload $WL > $F 2> $OUT/bench_${WL}_$MODE.err; fi
    echo "== bench $WL [$MODE: $E]" | tee -a $OUT/log.txt
    summ $F 2>&1 | tee -a $OUT/log.txt || tail -5 $OUT/bench_${WL}_$MODE.err | tee -a $OUT/log.txt
  done
done
echo "== done" | tee -a $OUT/log.txt

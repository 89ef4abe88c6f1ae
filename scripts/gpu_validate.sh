#!/bin/bash
# validation visit for one change: the named test files (fail fast), then the default workload's bench line without side measurements.
# usage: scripts/gpu_validate.sh TAG "tests/a.py tests/b.py" [pytest -k expression]
TAG=${1:-v}; FILES=${2:-tests}; KEXPR=${3:-}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then timeout 1500 python -m pytest $FILES -m gpu -x -q --durations=8 -k "$KEXPR" > $OUT/tests.txt 2>&1
else timeout 1500 python -m pytest $FILES -m gpu -x -q --durations=8 > $OUT/tests.txt 2>&1; fi
tail -25 $OUT/tests.txt | tee $OUT/log.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --no-north-star > $OUT/b.json 2> $OUT/b.err
python - $OUT/b.json <<'PY' | tee -a $OUT/log.txt
import json, sys
d = json.load(open(sys.argv[1]))
print("%7.2f Gbp/s %8.3f ms/step | " % (d["value"], d["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()) + " | index %.2f s" % d["config"]["index_build_s"])
PY

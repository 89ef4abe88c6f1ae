#!/bin/bash
# visit: parity subset, same-box old/new, seed-table layout A/B at human scale, default bench line
TAG=${1:-vb}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
scripts/gpu_ab.sh $TAG "" subset
scripts/gpu_old_new.sh $TAG r11c_tree configs1
summ() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-34s %7.2f Gbp/s %8.3f ms/step | " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()) + " | index %.1f s" % d["config"]["index_build_s"])
PY
}
for WL in northstar configs4; do
  for L in line bucket16; do
    MM_SEED_LAYOUT=$L timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-path --workload $WL > $OUT/b_${WL}_$L.json 2> $OUT/b_${WL}_$L.err
    summ "$WL $L" $OUT/b_${WL}_$L.json 2>&1 | tee -a $OUT/log.txt
  done
done
timeout 1200 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err | tee -a $OUT/log.txt
python - $OUT/bench.json <<'PY' | tee -a $OUT/log.txt
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["config"].get("host_synchronisations_per_pass"), d["config"]["index_build_s"])
print(json.dumps(d["roofline"]["kernels"])[:1500])
ns = d.get("north_star_target", {})
print({k: ns.get(k) for k in ("value", "ms_per_step", "index_build_s")}, json.dumps(ns.get("cpu_baseline"))[:600])
print(json.dumps(d.get("cpu_baseline"))[:300])
PY

#!/bin/bash
# MM_SKETCH_PROBE diagnostics: do k_sketch_fast (stream A) and k_seed_probe (stream B) overlap?  kernel trace with timestamps + bench variants
TAG=${1:-pd}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-north-star"
NS="--workload northstar"
for V in "MM_SKETCH_PROBE=0" "MM_SKETCH_PROBE=0 MM_NO_STEADY=1" "MM_SKETCH_PROBE=1" "MM_SKETCH_PROBE=1 MM_PROBE_PRIO=1" "MM_SKETCH_PROBE=1 MM_PROBE_CHUNKS=16" "MM_SKETCH_PROBE=1 MM_PROBE_CHUNKS=4" "NS MM_SKETCH_PROBE=0" "NS MM_SKETCH_PROBE=0 MM_NO_STEADY=1" "NS MM_SKETCH_PROBE=1" "NS MM_SKETCH_PROBE=1 MM_PROBE_CHUNKS=16"; do
  X=""; E="$V"; case "$V" in NS*) X="$NS"; E="${V#NS }";; esac
  env $E timeout 600 $B $X > $OUT/b.json 2> $OUT/b.err
  python - "$V" $OUT/b.json <<'PY' | tee -a $OUT/log.txt
import json, sys
d = json.load(open(sys.argv[2]))
print("%-52s %7.2f Gbp/s %8.3f ms/step | " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s %.2f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()))
PY
done
env MM_SKETCH_PROBE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $B > /dev/null 2> $OUT/trace.err
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python - $F <<'PY' | tee -a $OUT/log.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ks = [r for r in rows if "k_sketch_fast" in r["Kernel_Name"] or "k_seed_probe" in r["Kernel_Name"] or "k_lookup_l1" in r["Kernel_Name"] or "k_sketch_hard" in r["Kernel_Name"]]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(ks[0]["Start_Timestamp"])
for r in ks[-60:]:
    print("%-28s q%-3s start %10.3f ms dur %8.3f ms grid %s" % (r["Kernel_Name"][:28], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
PY
rm -rf $OUT/trace

#!/bin/bash
# Per-instruction view of one map kernel: rocprofv3 PC sampling (beta) over one pass of a workload, reduced to a histogram of sampled
# program counters of the kernel whose name matches KERNEL (default k_l2_locate), with the disassembly line of each hot address.
# Prepared in round 4 for the next one (not run: the round's GPU minutes were spent).  usage: scripts/gpu_pc_sample.sh TAG [KERNEL] [bench args...]
TAG=${1:-pcs}; KERNEL=${2:-k_l2_locate}; shift 2 2>/dev/null; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
# host_trap + time is the method every gfx9 part supports; stochastic sampling (cycles, with stall reasons) only where the hardware has it
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 \
  --kernel-trace --output-format csv json -d $OUT/pcs -o pcs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-north-star "$@" > /dev/null 2> $OUT/pcs.err
ls -la $OUT/pcs/* 2>/dev/null | head -20 | tee $OUT/log.txt
python - $OUT "$KERNEL" <<'PY' | tee -a $OUT/log.txt
import csv, glob, json, os, sys, collections
out, kern = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(out, "pcs", "**", "*pc_sampling*.csv"), recursive=True)
print("pc sampling files:", files)
hist = collections.Counter(); total = 0
for fn in files:
    with open(fn, newline="") as f:
        rd = csv.DictReader(f)
        cols = rd.fieldnames
        for row in rd:
            total += 1
            name = row.get("Kernel_Name") or row.get("kernel_name") or ""
            if kern not in name and name: continue
            key = row.get("Instruction") or row.get("instruction") or row.get("PC") or row.get("pc") or row.get("Code_Object_Offset") or str(sorted(row.items())[:3])
            hist[key] += 1
print("samples:", total, "columns:", cols if files else None)
n = sum(hist.values())
for key, c in hist.most_common(60):
    print("%6d %5.1f%%  %s" % (c, 100.0 * c / max(1, n), key[:150]))
PY

#!/bin/bash
# PMC pass over bench.py (1 step).  usage: scripts/gpu_pmc.sh TAG "COUNTER1 COUNTER2 ..." [reads]
TAG=$1; CTRS=$2; READS=${3:-1000000}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/pmc -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --reads $READS > /dev/null 2> $OUT/pmc.err
for C in $CTRS; do python scripts/pmc_summary.py $OUT/pmc $C | head -8 > $OUT/pmc_$C.csv; echo "== $C"; cat $OUT/pmc_$C.csv; done
rm -rf $OUT/pmc

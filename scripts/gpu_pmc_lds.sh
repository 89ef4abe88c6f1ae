#!/bin/bash
# LDS / issue counters of the hash loop (k_hash_only, k_sketch_fast) on the default workload
TAG=$1; cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc_$N -o pmc -- python bench.py --steps 1 --warmup 0 --reads 400000 --no-cpu-baseline --no-host-path > /dev/null 2> $OUT/pmc_$N.err
  for C in $SET; do python scripts/pmc_summary.py $OUT/pmc_$N $C 2>/dev/null | grep -E "k_hash_only|k_sketch_fast|k_l2_sweep|k_l2_locate|k_lookup" | sed "s/^/$C: /" | tee -a $OUT/lds.txt; done
  rm -rf $OUT/pmc_$N
done

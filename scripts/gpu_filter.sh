#!/bin/bash
# presence-filter size vs k_lookup_l1 time and fabric traffic on the default workload (configs[1], 100 Mbp)
TAG=$1; cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for B in 4 8 16 0; do
  MM_FILTER_BITS_PER_KEY=$B timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-path 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('filter bits/key >= $B:', d['value'], 'Gbp/s; lookup', round(d['kernels']['lookup']['ms_per_step'],3), 'ms; step', d['ms_per_step'])" | tee -a $OUT/filter.txt
done
for B in 4 16; do
  MM_FILTER_BITS_PER_KEY=$B timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_$B -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path > /dev/null 2> $OUT/pmc_$B.err
  python scripts/pmc_summary.py $OUT/pmc_$B FETCH_SIZE 2>/dev/null | grep -i "lookup\|kernel" | head -3 | sed "s/^/bits>=$B: /" | tee -a $OUT/filter.txt
  rm -rf $OUT/pmc_$B
done

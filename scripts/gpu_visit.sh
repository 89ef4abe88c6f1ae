#!/bin/bash
# One GPU-box visit.  usage: scripts/gpu_visit.sh TAG [steps...]   steps: tests probe bench small34 trace pmc
TAG=${1:-r02x}; shift
STEPS=${@:-tests bench}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for S in $STEPS; do
case $S in
tests)
  echo "== pytest -m gpu" | tee -a $OUT/log.txt
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --timeout 900 2>&1 | tail -60 | tee -a $OUT/log.txt ;;
smoke)
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee -a $OUT/log.txt ;;
probe)
  echo "== valu_rate probe" | tee -a $OUT/log.txt
  timeout 300 scripts/probes/valu_rate.bin > $OUT/valu_rate.txt 2>&1; tail -60 $OUT/valu_rate.txt | tee -a $OUT/log.txt ;;
bench)
  echo "== bench" | tee -a $OUT/log.txt
  MM_DEBUG=1 timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
  grep -v "^\[mm\] sketch" $OUT/bench.err | tail -12 | tee -a $OUT/log.txt; cat $OUT/bench.json | tee -a $OUT/log.txt ;;
small34)
  echo "== configs3 / configs4 scaled down" | tee -a $OUT/log.txt
  timeout 900 python bench.py --steps 3 --warmup 1 --workload configs3 --reads 100000 --ref-contigs 3 --no-cpu-baseline > $OUT/bench_c3s.json 2> $OUT/bench_c3s.err
  tail -4 $OUT/bench_c3s.err | tee -a $OUT/log.txt; cat $OUT/bench_c3s.json | tee -a $OUT/log.txt
  timeout 900 python bench.py --steps 3 --warmup 1 --workload configs4 --reads 60000 --ref-contigs 2 --ref-contig-len 150000000 --no-cpu-baseline > $OUT/bench_c4s.json 2> $OUT/bench_c4s.err
  tail -4 $OUT/bench_c4s.err | tee -a $OUT/log.txt; cat $OUT/bench_c4s.json | tee -a $OUT/log.txt ;;
c3)
  echo "== configs3 full" | tee -a $OUT/log.txt
  MM_DEBUG=1 timeout 1500 python bench.py --steps 3 --warmup 1 --workload configs3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
  grep -v "^\[mm\] sketch\|lookup+L1" $OUT/bench_c3.err | tail -12 | tee -a $OUT/log.txt; cat $OUT/bench_c3.json | tee -a $OUT/log.txt ;;
ns)
  echo "== north_star target: 10 kbp reads vs 3 Gbp" | tee -a $OUT/log.txt
  MM_DEBUG=1 timeout 1500 python bench.py --steps 3 --warmup 1 --workload northstar > $OUT/bench_ns.json 2> $OUT/bench_ns.err
  grep "index" $OUT/bench_ns.err | tail -8 | tee -a $OUT/log.txt; cat $OUT/bench_ns.json | tee -a $OUT/log.txt ;;
c4)
  echo "== configs4 full" | tee -a $OUT/log.txt
  MM_DEBUG=1 timeout 1500 python bench.py --steps 3 --warmup 1 --workload configs4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
  grep -v "^\[mm\] sketch\|lookup+L1" $OUT/bench_c4.err | tail -12 | tee -a $OUT/log.txt; cat $OUT/bench_c4.json | tee -a $OUT/log.txt ;;
occ)
  echo "== hash-only kernel vs LDS claimed per workgroup (occupancy)" | tee -a $OUT/log.txt
  for L in 0 27000 45000 78000; do
    MM_HASH_ONLY_LDS=$L timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --reads 500000 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds', $L, 'hash_only_ms', d['roofline']['int']['hash_only_ms'], 'sketch_ms', d['kernels']['sketch']['ms_per_step'])" | tee -a $OUT/log.txt
  done ;;
filt)
  echo "== presence filter vs 3 Gbp index (configs1 reads)" | tee -a $OUT/log.txt
  for CFG in "64 16" "160 4" "200 6" "250 8"; do
    set -- $CFG
    MM_FILTER_MAX_MIB=$1 MM_FILTER_BITS_PER_KEY=$2 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --reads 500000 --ref-contigs 30 --ref-contig-len 100000000 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('filter max MiB $1 bits/key $2:', d['value'], 'Gbp/s; lookup', d['kernels']['lookup']['ms_per_step'], 'ms; index', d['config']['index_build_s'], 's')" | tee -a $OUT/log.txt
  done ;;
prof3|prof4)
  WL=configs${S#prof}
  echo "== $WL: kernel trace + PMC passes" | tee -a $OUT/log.txt
  timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -o trace -- python bench.py --steps 2 --warmup 1 --workload $WL --no-cpu-baseline --no-host-path > $OUT/trace_$WL.json 2> $OUT/trace_$WL.err
  find $OUT/trace_$WL -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$WL.csv
  rm -rf $OUT/trace_$WL
  grep -E '^"(void )?k_' $OUT/kernel_stats_$WL.csv | head -10 | cut -c1-150 | tee -a $OUT/log.txt
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout 1500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${WL}_$C -o pmc -- python bench.py --steps 1 --warmup 0 --workload $WL --no-cpu-baseline --no-host-path > /dev/null 2> $OUT/pmc_${WL}_$C.err
    python scripts/pmc_summary.py $OUT/pmc_${WL}_$C $C > $OUT/pmc_${WL}_$C.csv 2>> $OUT/log.txt
    rm -rf $OUT/pmc_${WL}_$C
    head -8 $OUT/pmc_${WL}_$C.csv | tee -a $OUT/log.txt
  done ;;
scale)
  echo "== configs[2]: 3 Gbp assembly vs 3 Gbp reference, one-to-one, PAF vs the stock binary" | tee -a $OUT/log.txt
  NC=24 CL=125000000 THREADS=64 timeout 2400 python scripts/scale_probe.py 2>&1 | tail -25 | tee $OUT/scale_probe.txt | tee -a $OUT/log.txt ;;
trace)
  echo "== rocprofv3 kernel trace" | tee -a $OUT/log.txt
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path > $OUT/trace_bench.json 2> $OUT/trace.err
  find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
  rm -rf $OUT/trace
  head -14 $OUT/kernel_stats.csv | cut -c1-200 | tee -a $OUT/log.txt ;;
pmc)
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    echo "== pmc $C" | tee -a $OUT/log.txt
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path > /dev/null 2> $OUT/pmc_$C.err
    python scripts/pmc_summary.py $OUT/pmc_$C $C > $OUT/pmc_$C.csv 2>> $OUT/log.txt
    rm -rf $OUT/pmc_$C
    cat $OUT/pmc_$C.csv | tee -a $OUT/log.txt
  done ;;
esac
done
echo "== done" | tee -a $OUT/log.txt

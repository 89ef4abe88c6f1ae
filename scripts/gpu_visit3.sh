#!/bin/bash
# One GPU-box visit (round 3).  usage: scripts/gpu_visit3.sh TAG [steps...]
#   tests bench e2e prof:<workload> (kernel trace + FETCH/WRITE/VALU passes of bench.py --workload W) fuzz ...
TAG=${1:-r04x}; shift
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
tests:*)
  echo "== pytest ${S#tests:}" | tee -a $OUT/log.txt
  timeout 1200 python -m pytest ${S#tests:} -m gpu -q --maxfail=8 --timeout 900 2>&1 | tail -40 | tee -a $OUT/log.txt ;;
smoke)
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee -a $OUT/log.txt ;;
bench)
  echo "== bench (default line, north_star_target attached)" | tee -a $OUT/log.txt
  MM_DEBUG=1 timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
  grep -v "^\[mm\] sketch\|lookup+L1" $OUT/bench.err | tail -14 | tee -a $OUT/log.txt; cat $OUT/bench.json | tee -a $OUT/log.txt ;;
bench:*)
  WL=${S#bench:}
  echo "== bench --workload $WL" | tee -a $OUT/log.txt
  MM_DEBUG=1 timeout 1500 python bench.py --steps 3 --warmup 1 --workload $WL --no-cpu-baseline > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  grep -v "^\[mm\] sketch\|lookup+L1" $OUT/bench_$WL.err | tail -8 | tee -a $OUT/log.txt; cat $OUT/bench_$WL.json | tee -a $OUT/log.txt ;;
e2e)
  echo "== FASTA -> PAF end to end" | tee -a $OUT/log.txt
  for RT in ${E2E_THREADS:-8 12 16}; do
    MASHMAP_HIP_READER_THREADS=$RT timeout 900 python scripts/e2e_fasta_paf.py ${E2E_ARGS} > $OUT/e2e_rt$RT.json 2> $OUT/e2e_rt$RT.err
    tail -1 $OUT/e2e_rt$RT.json | tee -a $OUT/log.txt
    E2E_ARGS="--reuse"
  done
  MASHMAP_HIP_ASCII_UPLOAD=1 timeout 900 python scripts/e2e_fasta_paf.py --reuse > $OUT/e2e_ascii.json 2> $OUT/e2e_ascii.err; tail -1 $OUT/e2e_ascii.json | tee -a $OUT/log.txt ;;
prof:*)
  WL=${S#prof:}
  echo "== $WL: kernel trace + PMC passes" | tee -a $OUT/log.txt
  timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -o trace -- python bench.py --steps 2 --warmup 1 --workload $WL --no-cpu-baseline --no-host-path --no-north-star > $OUT/trace_$WL.json 2> $OUT/trace_$WL.err
  find $OUT/trace_$WL -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$WL.csv
  rm -rf $OUT/trace_$WL
  grep -E '^"(void )?k_' $OUT/kernel_stats_$WL.csv | head -10 | cut -c1-150 | tee -a $OUT/log.txt
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout 1500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${WL}_$C -o pmc -- python bench.py --steps 1 --warmup 0 --workload $WL --no-cpu-baseline --no-host-path --no-north-star > /dev/null 2> $OUT/pmc_${WL}_$C.err
    python scripts/pmc_summary.py $OUT/pmc_${WL}_$C $C > $OUT/pmc_${WL}_$C.csv 2>> $OUT/log.txt
    rm -rf $OUT/pmc_${WL}_$C
    head -8 $OUT/pmc_${WL}_$C.csv | tee -a $OUT/log.txt
  done ;;
fuzz)
  echo "== fuzz parity" | tee -a $OUT/log.txt
  timeout 900 python scripts/fuzz_parity.py ${FUZZ_N:-20} ${FUZZ_SEED:-5} 2>&1 | tail -25 | tee -a $OUT/log.txt ;;
fuzzpaf)
  echo "== fuzz PAF" | tee -a $OUT/log.txt
  timeout 900 python scripts/fuzz_paf.py ${FUZZ_N:-15} ${FUZZ_SEED:-5} 2>&1 | tail -20 | tee -a $OUT/log.txt ;;
esac
done
echo "== done" | tee -a $OUT/log.txt

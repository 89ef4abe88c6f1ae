#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel trace, PMC passes.  usage: scripts/gpu_round.sh TAG [quick]
TAG=${1:-r01x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
echo "== pytest -m gpu" | tee $OUT/log.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee -a $OUT/log.txt
echo "== smoke" | tee -a $OUT/log.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee -a $OUT/log.txt
echo "== bench" | tee -a $OUT/log.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/bench.err | tee -a $OUT/log.txt; cat $OUT/bench.json | tee -a $OUT/log.txt
if [ "$2" != "quick" ]; then
echo "== rocprofv3 kernel trace" | tee -a $OUT/log.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -f $(find $OUT/trace -name '*kernel_trace.csv')
head -12 $OUT/kernel_stats.csv | cut -c1-200 | tee -a $OUT/log.txt
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  echo "== pmc $C" | tee -a $OUT/log.txt
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$C.err
  python scripts/pmc_summary.py $OUT/pmc_$C $C > $OUT/pmc_$C.csv 2>> $OUT/log.txt
  rm -rf $OUT/pmc_$C
  cat $OUT/pmc_$C.csv | tee -a $OUT/log.txt
done
fi
rm -rf $OUT/trace
echo "== done" | tee -a $OUT/log.txt

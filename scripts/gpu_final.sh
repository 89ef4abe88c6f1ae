#!/bin/bash
# The round's closing visit: parity suite + smoke, the default bench line (with north_star_target and both CPU baselines), the other
# BASELINE shapes, and for every workload a rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes.
# usage: scripts/gpu_final.sh TAG [notests]
TAG=${1:-r12a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
: > $OUT/log.txt
if [ "$2" != notests ]; then
  echo "== pytest -m gpu" | tee -a $OUT/log.txt
  timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -20 | tee $OUT/gpu_tests.txt | tee -a $OUT/log.txt
  echo "== smoke" | tee -a $OUT/log.txt
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee -a $OUT/gpu_tests.txt | tee -a $OUT/log.txt
fi
prof() {   # name, bench args...
  NAME=$1; shift
  echo "== rocprofv3 kernel trace: $NAME" | tee -a $OUT/log.txt
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-north-star "$@" > /dev/null 2> $OUT/trace_$NAME.err
  find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/${NAME}_kernel_stats.csv
  rm -rf $OUT/trace
  head -9 $OUT/${NAME}_kernel_stats.csv | cut -c1-160 | tee -a $OUT/log.txt
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-north-star "$@" > /dev/null 2> $OUT/pmc_${NAME}_$C.err
    python scripts/pmc_summary.py $OUT/pmc_$C $C > $OUT/${NAME}_pmc_$C.csv 2>> $OUT/log.txt
    cp $OUT/${NAME}_pmc_$C.csv profiles/${TAG}_${NAME}_pmc_$C.csv
    rm -rf $OUT/pmc_$C
    head -6 $OUT/${NAME}_pmc_$C.csv | cut -c1-120 | tee -a $OUT/log.txt
  done
  cp $OUT/${NAME}_kernel_stats.csv profiles/${TAG}_${NAME}_kernel_stats.csv
  python scripts/pmc_to_json.py ${TAG}_${NAME} $NAME > /dev/null 2>> $OUT/log.txt     # profiles/pmc_traffic.json[NAME]: what the bench lines below cite
}
prof configs1
prof northstar --workload northstar
prof northstar_seg10000 --workload northstar --seg 10000
prof configs3 --workload configs3
prof configs4 --workload configs4
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
echo "== bench (default)" | tee -a $OUT/log.txt
timeout 1500 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -6 $OUT/bench.err | tee -a $OUT/log.txt; cut -c1-600 $OUT/bench.json | tee -a $OUT/log.txt
for WL in configs3 configs4; do
  timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  cut -c1-300 $OUT/bench_$WL.json | tee -a $OUT/log.txt
done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-path --workload northstar --seg 10000 > $OUT/bench_northstar_seg10000.json 2> /dev/null
echo "== done" | tee -a $OUT/log.txt

#!/bin/bash
# One visit to a GPU box (gpurun): scripts/gpu.sh TAG STEP [STEP ...]; everything lands under gpurun_out/TAG/ (log.txt has the gist).
#   tests            pytest -m gpu (the whole suite)             t:PATTERN   pytest -m gpu -k PATTERN
#   smoke            __graft_entry__.smoke()
#   bench            the default bench line (configs[1] + e2e + north_star_target incl. repeat_rich + configs2), steps 20 / warmup 5; the full
#                    record of every bench run is kept as <step>_full.json and digested (the last stdout line is the short one the driver parses)
#   bench:WL         bench.py --workload WL (configs3 | configs4 | northstar), steps 5 / warmup 3, no side measurements
#   c2               bench.py --workload configs2: resident passes + FASTA -> PAF of the 3 Gbp assembly (C2_EXTRA=--stock: the stock binary beside it)
#   quick            configs[1] headline only (no cpu baseline / e2e / north star), steps 10 / warmup 3
#   rr               the repeat-rich north_star workload alone (bench.py --workload northstar --repeat-rich-reference)
#   trace:WL         rocprofv3 --kernel-trace --stats of two passes of WL      -> kernel_stats_WL.csv
#   pmc:WL           FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes of WL (one rocprofv3 run each) -> pmc_WL_*.csv (scripts/pmc_summary.py reduces them);
#                    WL: configs1 | configs2 | configs3 | configs4 | northstar | northstar_seg10000 | northstar_repeat_rich (also for trace:)
#   e2e              FASTA -> PAF through the mashmap_hip command line only (bench.py's e2e leg)
#   fuzz:N           N random command lines through mashmap_hip and the stock binary, PAF bytes compared (MM_FUZZ_SEED)
#   nosplit          --noSplit on 30 kbp reads through both command lines: same PAF, both times
#   large            --dense -s 100000 (sketchSize 9 998) and k = 40 / 57 through mashmap_hip and the stock binary, with the stage log
# Environment variables given on the command line reach every step (A/B switches: MM_*, MASHMAP_HIP_*); MM_BENCH_EXTRA: extra bench.py
# arguments of the trace steps (e.g. --repeat-rich-reference).
TAG=${1:-visit}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
say() { echo "$@" | tee -a $OUT/log.txt; }
# bench.py prints the short line the driver parses and writes the full record to gpurun_out/bench_last_full.json: keep it per run, digest that
full() { cp gpurun_out/bench_last_full.json $OUT/$1_full.json 2>/dev/null && python scripts/bench_digest.py $OUT/$1_full.json | tee -a $OUT/log.txt; tail -c 6000 $OUT/$1.json | tail -1 | wc -c | sed 's/^/   final line bytes: /' | tee -a $OUT/log.txt; }
wl_args() { case $1 in configs1|"") echo "";; northstar_seg10000) echo "--workload northstar --seg 10000";; northstar_repeat_rich) echo "--workload northstar --repeat-rich-reference";; *) echo "--workload $1";; esac; }
for S in "$@"; do
case $S in
tests)
  say "== pytest -m gpu"
  timeout 1700 python -m pytest tests -m gpu -q --maxfail=12 --timeout 900 -s --durations=12 2>&1 | grep -v "^$" | tail -80 | tee -a $OUT/log.txt ;;
t:*)
  say "== pytest -m gpu -k ${S#t:}"
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --timeout 900 -k "${S#t:}" 2>&1 | tail -60 | tee -a $OUT/log.txt ;;
smoke)
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee -a $OUT/log.txt ;;
bench)
  say "== bench (default line)"
  MM_E2E_LOG=$OUT/e2e_stage_log.txt timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
  grep -v "^\[mm\]" $OUT/bench.err | tail -25 | tee -a $OUT/log.txt; full bench ;;
bench:*)
  WL=${S#bench:}
  say "== bench --workload $WL"
  timeout 1500 python bench.py --steps 5 --warmup 3 --workload $WL --no-cpu-baseline --no-host-path > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  tail -5 $OUT/bench_$WL.err | tee -a $OUT/log.txt; full bench_$WL ;;
c2)
  say "== bench --workload configs2 (resident passes + FASTA -> PAF through the command line; C2_EXTRA=--stock adds the stock binary)"
  MM_E2E2_LOG=$OUT/e2e_configs2_stage_log.txt timeout 1700 python bench.py --steps 5 --warmup 3 --workload configs2 --no-cpu-baseline $C2_EXTRA > $OUT/bench_configs2.json 2> $OUT/bench_configs2.err
  grep -v "^\[mm\]" $OUT/bench_configs2.err | tail -8 | cut -c1-1200 | tee -a $OUT/log.txt; cat $OUT/e2e_configs2_stage_log.txt | tee -a $OUT/log.txt; full bench_configs2 ;;
quick)
  say "== configs[1] headline only"
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-north-star > $OUT/quick.json 2> $OUT/quick.err
  tail -4 $OUT/quick.err | tee -a $OUT/log.txt; full quick ;;
rr)
  say "== repeat-rich north_star workload"
  MM_DEBUG=${MM_DEBUG:-} timeout 1500 python bench.py --steps 5 --warmup 3 --workload northstar --repeat-rich-reference --no-cpu-baseline --no-host-path > $OUT/rr.json 2> $OUT/rr.err
  grep -v "^\[mm\] sketch" $OUT/rr.err | tail -12 | tee -a $OUT/log.txt; full rr ;;
trace:*)
  WL=${S#trace:}
  say "== $WL: rocprofv3 --kernel-trace --stats"
  timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -o trace -- python bench.py --steps 3 --warmup 3 $(wl_args $WL) $MM_BENCH_EXTRA --no-cpu-baseline --no-host-path --no-e2e --no-north-star > $OUT/trace_$WL.json 2> $OUT/trace_$WL.err
  find $OUT/trace_$WL -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$WL.csv
  rm -rf $OUT/trace_$WL
  grep -E '^"(void )?k_' $OUT/kernel_stats_$WL.csv | head -12 | cut -c1-160 | tee -a $OUT/log.txt ;;
pmc:*)
  WL=${S#pmc:}
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    say "== $WL: rocprofv3 --pmc $C"
    timeout 1500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${WL}_$C -o pmc -- python bench.py --steps 1 --warmup 3 --batches 1 $(wl_args $WL) $MM_BENCH_EXTRA --no-cpu-baseline --no-host-path --no-e2e --no-north-star > /dev/null 2> $OUT/pmc_${WL}_$C.err
    python scripts/pmc_summary.py $OUT/pmc_${WL}_$C $C > $OUT/pmc_${WL}_$C.csv 2>> $OUT/log.txt; rm -rf $OUT/pmc_${WL}_$C
    head -8 $OUT/pmc_${WL}_$C.csv | cut -c1-160 | tee -a $OUT/log.txt
  done ;;
e2e)
  say "== e2e FASTA -> PAF"
  MM_E2E_LOG=$OUT/e2e_stage_log.txt timeout 900 python -c "
import json, os, sys, torch
sys.path.insert(0, '.')
import bench as B
dev = torch.device('cuda', 0); W = dict(B.WORKLOADS['configs1'])
ref = B.contiguous_views(torch, B.make_reference(torch, dev, W['ref_contigs'], W['ref_contig_len']))
print(json.dumps(B.e2e_fasta_to_paf(torch, dev, W, ref, W['reads'], max(4, min(128, os.cpu_count())))))" > $OUT/e2e.json 2> $OUT/e2e.err
  tail -3 $OUT/e2e.err | tee -a $OUT/log.txt; cat $OUT/e2e.json | cut -c1-2500 | tee -a $OUT/log.txt ;;
large)
  say "== large sketches / long k-mers through both command lines (scripts/large_sketch_paf.py)"
  LARGE_VERBOSE=1 timeout 900 python scripts/large_sketch_paf.py 2>&1 | tail -60 | tee -a $OUT/log.txt ;;
nosplit)
  say "== --noSplit, 30 kbp reads, both command lines (scripts/nosplit_paf.py)"
  timeout 900 python scripts/nosplit_paf.py 2>&1 | tail -8 | tee -a $OUT/log.txt ;;
fuzz:*)
  say "== PAF fuzz, ${S#fuzz:} random command lines through both programs (scripts/fuzz_paf.py)"
  timeout 1500 python scripts/fuzz_paf.py ${S#fuzz:} ${MM_FUZZ_SEED:-5} > $OUT/fuzz_paf.txt 2>&1; grep -c "^ok" $OUT/fuzz_paf.txt | tee -a $OUT/log.txt; grep -A6 "^FAIL" $OUT/fuzz_paf.txt | head -40 | tee -a $OUT/log.txt; tail -1 $OUT/fuzz_paf.txt | tee -a $OUT/log.txt ;;
*) say "unknown step $S" ;;
esac
done

#!/bin/bash
# sketch kernel time vs the geometry of the fast kernel: MM_SKETCH_CUT = wanted survivors / s, MM_SKETCH_HTF = table slots per wanted survivor,
# MM_SKETCH_QSIG = standard deviations of head room in the per-wave queues.  usage: scripts/gpu_cut.sh TAG
TAG=${1:-r02x}; OUT=gpurun_out/$TAG; mkdir -p $OUT
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
run() {  # workload "cut htf qsig"...
  WL=$1; shift
  for CFG in "$@"; do
    set -- $CFG
    unset MM_SKETCH_CUT MM_SKETCH_HTF MM_SKETCH_QSIG
    [ "$1" != d ] && export MM_SKETCH_CUT=$1; [ "$2" != d ] && export MM_SKETCH_HTF=$2; [ "$3" != d ] && export MM_SKETCH_QSIG=$3
    MM_DEBUG=1 timeout 600 python bench.py --steps 3 --warmup 1 --workload $WL --reads 400000 --ref-contigs 2 --ref-contig-len 50000000 --no-cpu-baseline --no-host-path 2> $OUT/err.txt |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$WL s', d['config']['sketchSize'], 'cut/htf/qsig $CFG: sketch %.2f ms + hard %.2f ms; hash-only %.2f ms; step %.2f ms' % (k['sketch']['ms_per_step'], k['sketch_hard']['ms_per_step'], d['roofline']['int']['hash_only_ms'], d['ms_per_step']))" | tee -a $OUT/log.txt
    grep "to the hard path" $OUT/err.txt | tail -1 | tee -a $OUT/log.txt
  done
}
run configs1 "d d d" "1.37 d d" "1.30 d d" "1.30 2.2 5" "1.26 2.2 5" "1.26 2.0 5"
run configs3 "d d d" "1.24 d d" "1.24 2.2 d" "1.24 2.2 5" "1.20 2.2 5" "1.20 2.0 5"
run configs4 "d d d" "1.19 d d" "1.19 2.2 d" "1.19 2.2 5" "1.16 2.2 5" "1.16 2.0 5"

#!/bin/bash
# A/B of several builds of libmashmap_hip.so on ONE GPU box (boxes differ by a few per cent): scripts/ab_libs.sh TAG WORKLOAD V1 V2 ...
# Each variant is gpurun_aux/<V>/libmashmap_hip.so (gpurun_aux/ is git-ignored but travels with gpurun); build one from another tree's
# source with the Makefile's flags, e.g.
#   git show REV:mashmap_amd/csrc/mm_l2.hip > mashmap_amd/csrc/x.hip; hipcc <FLAGS> -c mashmap_amd/csrc/x.hip -o x.o; hipcc -shared -o gpurun_aux/P/libmashmap_hip.so <other objects> x.o
# The variants run in the order given (repeat a name to interleave); the library in mashmap_amd/lib/ is put back at the end.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
TAG=$1; WL=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp mashmap_amd/lib/libmashmap_hip.so /tmp/libmashmap_hip.keep
WLARG=""; [ "$WL" != configs1 ] && WLARG="--workload $WL"
i=0
for V in "$@"; do
  i=$((i+1))
  cp gpurun_aux/$V/libmashmap_hip.so mashmap_amd/lib/libmashmap_hip.so
  timeout 600 python bench.py --steps 6 --warmup 3 $WLARG --no-cpu-baseline --no-host-path --no-e2e --no-north-star > $OUT/${i}_$V.json 2> $OUT/${i}_$V.err
  echo "== $V $WL" | tee -a $OUT/log.txt; python scripts/bench_digest.py $OUT/${i}_$V.json | sed -n 2,3p | tee -a $OUT/log.txt
done
cp /tmp/libmashmap_hip.keep mashmap_amd/lib/libmashmap_hip.so

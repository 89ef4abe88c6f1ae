#!/bin/bash
# sketch-kernel timing (and, with MM_SKETCH_STATS, its phases) at the three BASELINE sketch sizes; small reference: the sketch kernel does not depend on it
TAG=$1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/$TAG
for W in "configs1 400000" "configs3 250000" "configs4 200000"; do set -- $W
  WL=$1; NR=$2
  timeout 600 python bench.py --steps 3 --warmup 1 --workload $WL --reads $NR --ref-contigs 1 --ref-contig-len 50000000 --no-cpu-baseline --no-host-path 2>gpurun_out/$TAG/stats_$WL.err >gpurun_out/$TAG/stats_$WL.json
  python -c "import json;d=json.load(open('gpurun_out/$TAG/stats_$WL.json'));k=d['kernels'];print('$WL', d['value'],'Gbp/s; per 1M fragments: sketch', round(k['sketch']['ms_per_step']/d['config']['fragments_per_gpu']*1e6,2),'ms, hash-only', round(d['roofline']['int']['hash_only_ms']/d['config']['fragments_per_gpu']*1e6,2), 'ms, frac', d['roofline']['int']['sketch_kernel_frac'])" | tee -a gpurun_out/$TAG/phases.txt
  MM_SKETCH_STATS=1 timeout 600 python bench.py --steps 1 --warmup 0 --workload $WL --reads $NR --ref-contigs 1 --ref-contig-len 50000000 --no-cpu-baseline --no-host-path 2>&1 >/dev/null | grep "sketch phases" | tail -1 | tee -a gpurun_out/$TAG/phases.txt
done

"""One-off randomized parity campaign on the GPU: random (k, segLength, sketchSize, pi, flags, error rate, genome shape) scenarios
(tests/gpucheck.py::fuzz_scenario) through the C ABI vs the CPU oracle, every integer of every stage.
usage: fuzz_parity.py [n_iter] [seed0]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mmutil as U
from gpucheck import run_and_compare, fuzz_scenario

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = U.Oracle()
bad = 0
t00 = time.time()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_index import _compare as index_compare
for it in range(n_iter):
    contigs, reads, kw, desc = fuzz_scenario(seed0, it)
    try:
        t0 = time.time()
        nF, nl = run_and_compare(orc, contigs, reads, verbose=True, **kw)
        nm, nf = index_compare(orc, contigs, kw["k"], kw["L"], kw["s"], kw["kmerPct"])      # mm_index_build (device winnowing) vs the oracle's index
        print("ok  ", desc, "frags", nF, "loci", nl, "minmers", nm, "%.1fs" % (time.time() - t0), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", desc, "\n   ", str(e)[:600], flush=True)
        traceback.print_exc(limit=2)
print("fuzz done: %d iterations, %d failures, %.0f s" % (n_iter, bad, time.time() - t00))
sys.exit(1 if bad else 0)

"""--noSplit (a read longer than segLength is ONE fragment, windowLen = len - segLength != 0: the literal kernels k_l1_window / k_l2_window)
through both command lines on 30 kbp reads: the PAF files must be byte-identical; prints both programs' times ('time spent mapping the
query' and wall).  usage: nosplit_paf.py [reads] [reference Mbp]"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mmutil as U

HIP = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mbp = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
td = tempfile.mkdtemp()
cs = [U.random_dna(7100 + i, int(mbp * 1e6 / 4)) for i in range(4)]
rf = os.path.join(td, "ref.fa"); U.write_fasta(rf, [("chr%d" % i, c) for i, c in enumerate(cs)])
reads = [("r%d" % i, a) for i, (_, a, _) in enumerate(U.sample_reads(cs, 7200, nreads, 30000, 0.10))]
qf = os.path.join(td, "q.fa"); U.write_fasta(qf, reads)
bad = 0
for args in (["--noSplit"], ["--noSplit", "--pi", "90", "-n", "3"]):
    outs = {}
    for tag, exe in (("hip", HIP), ("ref", U.REF_BIN)):
        t0 = time.time()
        p = subprocess.run([exe, "-r", rf, "-q", qf, "-t", "16", "-o", os.path.join(td, tag + ".paf")] + args, capture_output=True, text=True)
        m = re.search(r"time spent mapping the query\s*:\s*([0-9.eE+-]+)", p.stderr)
        outs[tag] = (p.returncode, open(os.path.join(td, tag + ".paf"), "rb").read() if p.returncode == 0 else p.stderr[-400:], time.time() - t0, float(m.group(1)) if m else -1.0)
    ok = outs["hip"][0] == 0 and outs["ref"][0] == 0 and outs["hip"][1] == outs["ref"][1]
    bad += 0 if ok else 1
    print("ok  " if ok else "FAIL", " ".join(args), "| %d reads x 30 kbp vs %.0f Mbp, %d PAF lines | mapping: hip %.2f s, stock %.2f s (-t 16) | wall: hip %.1f s, stock %.1f s"
          % (nreads, mbp, outs["ref"][1].count(b"\n") if outs["ref"][0] == 0 else -1, outs["hip"][3], outs["ref"][3], outs["hip"][2], outs["ref"][2]), flush=True)
    if not ok: print("   rc", outs["hip"][0], outs["ref"][0], str(outs["hip"][1])[-300:] if outs["hip"][0] else "", flush=True)
print("--noSplit done: %d failures" % bad)
sys.exit(1 if bad else 0)

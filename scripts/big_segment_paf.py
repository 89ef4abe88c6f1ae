"""Segment lengths far beyond the defaults through both command lines (mashmap_hip and the stock binary built from the reference
sources): -s 100000 / 300000 with reads of several segments, split and --noSplit; the PAF files must be byte-identical.  A fragment of
that length does not fit a CU's LDS: the exact sketch kernel's stream mode and the L2 stream's skip entries carry these runs."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mmutil as U

HIP = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
td = tempfile.mkdtemp()
cs = [U.random_dna(9000 + i, n) for i, n in enumerate((2500000, 1800000, 900000))]
blk = U.mutate(cs[0][200000:900000], 5, 0.02); cs[1][300000:300000 + len(blk)] = blk          # a 700 kbp repeat between two contigs
rf = os.path.join(td, "ref.fa"); U.write_fasta(rf, [("chr%d" % i, c) for i, c in enumerate(cs)])
reads = [("long%d" % i, U.mutate(cs[i % 3][o:o + n], 40 + i, e)) for i, (o, n, e) in enumerate([(100000, 750000, 0.03), (50000, 320000, 0.08), (10000, 880000, 0.01), (400000, 610000, 0.05)])]
reads.append(("rc", U.revcomp(U.mutate(cs[0][1200000:1900000], 77, 0.04))))
reads.append(("short", cs[2][5000:45000].copy()))
qf = os.path.join(td, "q.fa"); U.write_fasta(qf, reads)
bad = 0
for args in (["-s", "100000", "--pi", "90"], ["-s", "300000", "--pi", "85"], ["-s", "100000", "--pi", "85", "--noSplit"], ["-s", "300000", "--pi", "90", "-f", "none", "-n", "2"],
             ["-s", "50000", "--pi", "80", "--dense"]):
    outs = {}
    for tag, exe in (("hip", HIP), ("ref", U.REF_BIN)):
        p = subprocess.run([exe, "-r", rf, "-q", qf, "-t", "8", "-o", os.path.join(td, tag + ".paf")] + args, capture_output=True, text=True)
        outs[tag] = (p.returncode, open(os.path.join(td, tag + ".paf"), "rb").read() if p.returncode == 0 else p.stderr[-400:])
    ok = outs["hip"][0] == 0 and outs["ref"][0] == 0 and outs["hip"][1] == outs["ref"][1]
    bad += 0 if ok else 1
    print("ok  " if ok else "FAIL", " ".join(args), "lines", outs["ref"][1].count(b"\n") if outs["ref"][0] == 0 else -1, flush=True)
    if not ok: print("   rc", outs["hip"][0], outs["ref"][0], str(outs["hip"][1])[-300:] if outs["hip"][0] else "", str(outs["ref"][1])[-200:] if outs["ref"][0] else "", flush=True)
print("big segments done: %d failures" % bad)
sys.exit(1 if bad else 0)

"""Parameter ranges that were refused until rounds 4 / 6, through both command lines (mashmap_hip and the stock binary built from the reference
sources): --dense --pi 80 -s 200000 (sketchSize 19 998: the index build's sketch in HBM too), -s 100000 (9 998: the global-memory sketch
kernel and the literal L2 kernel) and k-mers of more than
32 bases (-k 40 / 57); the PAF files must be byte-identical."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mmutil as U

HIP = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
td = tempfile.mkdtemp()
cs = [U.random_dna(9100 + i, n) for i, n in enumerate((1500000, 1100000, 600000))]
blk = U.mutate(cs[0][200000:600000], 5, 0.02); cs[1][300000:300000 + len(blk)] = blk
rf = os.path.join(td, "ref.fa"); U.write_fasta(rf, [("chr%d" % i, c) for i, c in enumerate(cs)])
reads = [("long%d" % i, U.mutate(cs[i % 3][o:o + n], 40 + i, e)) for i, (o, n, e) in enumerate([(100000, 450000, 0.03), (50000, 320000, 0.08), (10000, 380000, 0.01), (400000, 210000, 0.05)])]
reads.append(("rc", U.revcomp(U.mutate(cs[0][900000:1300000], 77, 0.04))))
reads.append(("short", cs[2][5000:45000].copy()))
qf = os.path.join(td, "q.fa"); U.write_fasta(qf, reads)
bad = 0
for args in (["-s", "200000", "--pi", "80", "--dense"], ["-s", "100000", "--pi", "80", "--dense"], ["-k", "40", "-s", "20000", "--pi", "95"], ["-k", "57", "-s", "10000", "--pi", "97", "-f", "none"]):
    outs = {}
    for tag, exe in (("hip", HIP), ("ref", U.REF_BIN)):
        t0 = time.time()
        p = subprocess.run([exe, "-r", rf, "-q", qf, "-t", "8", "-o", os.path.join(td, tag + ".paf")] + args, capture_output=True, text=True,
                           env=dict(os.environ, MASHMAP_HIP_TIMING="1", MM_DEBUG="1") if tag == "hip" and os.environ.get("LARGE_VERBOSE") else None)
        if tag == "hip" and os.environ.get("LARGE_VERBOSE"):
            print("\n".join(l[:220] for l in p.stderr.splitlines() if "timing" in l or "time spent" in l or l.startswith("[mm] index") or "L2" in l)[-6000:], flush=True)
        outs[tag] = (p.returncode, open(os.path.join(td, tag + ".paf"), "rb").read() if p.returncode == 0 else p.stderr[-400:], time.time() - t0)
    ok = outs["hip"][0] == 0 and outs["ref"][0] == 0 and outs["hip"][1] == outs["ref"][1]
    bad += 0 if ok else 1
    print("ok  " if ok else "FAIL", " ".join(args), "lines", outs["ref"][1].count(b"\n") if outs["ref"][0] == 0 else -1, "| hip %.1f s, stock %.1f s" % (outs["hip"][2], outs["ref"][2]), flush=True)
    if not ok: print("   rc", outs["hip"][0], outs["ref"][0], str(outs["hip"][1])[-300:] if outs["hip"][0] else "", str(outs["ref"][1])[-200:] if outs["ref"][0] else "", flush=True)
print("large sketches / long k-mers done: %d failures" % bad)
sys.exit(1 if bad else 0)

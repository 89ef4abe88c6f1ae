"""One-off end-to-end campaign on the GPU box: random command lines through mashmap_hip and the stock binary (oracle/_ref/mashmap_ref,
built from the reference sources); the PAF files must be byte-identical.  usage: fuzz_paf.py [n_iter] [seed0]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mmutil as U

HIP = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
td = tempfile.mkdtemp()
for it in range(n_iter):
    r = U.splitmix64(seed0 * 104729 + it, 32)
    pick = lambda i, xs: xs[int(r[i] % np.uint64(len(xs)))]
    L = pick(0, [1000, 2000, 5000, 5000, 10000])
    nct = pick(1, [1, 2, 4])
    cs = [U.random_dna(5000 * it + i + seed0 * 77, int(30 * L + int(r[2 + i]) % (40 * L))) for i in range(nct)]
    if nct > 1 and pick(6, [0, 1]):
        blk = U.mutate(cs[0][:6 * L], 11, 0.03); cs[1][L:L + len(blk)] = blk[:len(cs[1]) - L]
    if pick(7, [0, 0, 1]):
        cs[0] = U.with_n_runs(cs[0], it, 3, L // 2)
    if pick(24, [0, 0, 1]):                                 # a repeat family: reads out of it bring more interval points than the fused lookup holds (HBM point path, k_filter_points)
        unit = cs[-1][L:3 * L].copy()
        fam = np.concatenate([U.mutate(unit, 500 + j, 0.02) for j in range(int(pick(25, [6, 10, 16])))])
        cs[0] = np.concatenate([cs[0], fam])
    allvsall = pick(8, [0, 0, 0, 1])
    if allvsall:
        names = ["S%d#1#c%d" % (i % 2, i) for i in range(nct)]
        qrec = None
    else:
        names = ["chr%d" % i for i in range(nct)]
        rl = pick(9, [L, 2 * L, 3 * L + 17, 4 * L, L // 2 + 40])
        err = pick(10, [0.0, 0.03, 0.08, 0.12])
        qrec = [(n_, a) for n_, a, _ in U.sample_reads(cs, it + 9, 30, min(rl, min(len(c) for c in cs)), err)]
        qrec.append(("chimera", np.concatenate([cs[0][L:3 * L], U.revcomp(cs[-1][2 * L:4 * L])])))
    args = ["-s", str(L), "--pi", str(pick(11, [80, 85, 85, 90, 95]))]
    if pick(12, [0, 1]): args += ["-J", str(pick(13, [20, 50, 100, 130, 200]))]
    elif pick(12, [0, 0, 1]): args += ["--dense"]
    args += pick(14, [[], [], ["-n", "2"], ["-n", "3"]])
    args += pick(15, [[], [], ["-f", "one-to-one"], ["-f", "none"]])
    args += pick(16, [[], [], ["-M"], ["-K"], ["--noHgFilter"], ["--legacy"], ["--reportPercentage"], ["--filterLengthMismatches"]])
    args += pick(17, [[], [], ["-l", str(2 * L)], ["-c", str(3 * L)], ["--kmerThreshold", "0.5"], ["-k", "16"], ["--hgFilterAniDiff", "1"]])
    if allvsall: args += pick(18, [["-Y", "#"], ["-X"], ["-X", "--lowerTriangular"], []])
    args += pick(21, [[], [], [], ["--noSplit"]])           # reads longer than a segment as one fragment (windowLen != 0)
    if pick(22, [0, 0, 0, 0, 1]): args += ["--dense", "--pi", "80"] if "--dense" not in args and "-J" not in args else []
    rf = os.path.join(td, "r%d.fa" % it); U.write_fasta(rf, list(zip(names, cs)))
    base = ["-r", rf, "-t", "4"] + args
    if qrec is not None:
        qf = os.path.join(td, "q%d.fa" % it); U.write_fasta(qf, qrec); base += ["-q", qf]
    outs = {}
    shard = pick(19, ["", "", "0,0", "0,0,0"])              # some runs sharded over several contexts (MASHMAP_HIP_DEVICES)
    for tag, exe in (("hip", HIP), ("ref", U.REF_BIN)):
        env = dict(os.environ)
        if tag == "hip" and shard: env["MASHMAP_HIP_DEVICES"] = shard; env["MASHMAP_HIP_BATCH_MBP"] = pick(20, ["512", "0.04", "0.2"])
        if tag == "hip" and not shard:                       # one context: small reader batches, several of them per device pass (or none: COALESCE 0)
            env["MASHMAP_HIP_BATCH_MBP"] = pick(26, ["512", "0.03", "0.1", "0.01"]); env["MASHMAP_HIP_COALESCE_MBP"] = pick(27, ["2048", "0", "0.25", "1"])
        if tag == "hip" and pick(23, [0, 0, 1]): env["MM_SEED_TAGS"] = "1"                       # the human-scale seed table layout forced onto the small index
        if tag == "hip" and pick(23, [0, 1, 0]): env["MASHMAP_HIP_ASCII_UPLOAD"] = "1"           # the ASCII upload path instead of the packing parser
        p = subprocess.run([exe] + base + ["-o", os.path.join(td, tag + ".paf")], capture_output=True, text=True, env=env)
        outs[tag] = (p.returncode, open(os.path.join(td, tag + ".paf"), "rb").read() if p.returncode == 0 else p.stderr[-300:])
    ok = outs["hip"][0] == 0 and outs["ref"][0] == 0 and outs["hip"][1] == outs["ref"][1]
    if not ok: bad += 1
    nl = outs["ref"][1].count(b"\n") if outs["ref"][0] == 0 else -1
    print("ok  " if ok else "FAIL", it, " ".join(args), "allvsall" if allvsall else "reads", "lines", nl, ("devices " + shard) if shard else "", flush=True)
    if not ok:
        if outs["hip"][0] != 0 or outs["ref"][0] != 0: print("   rc", outs["hip"][0], outs["ref"][0], str(outs["hip"][1])[-200:] if outs["hip"][0] else "", flush=True)
        else:
            a, b = outs["hip"][1].decode().splitlines(), outs["ref"][1].decode().splitlines()
            for i, (x, y) in enumerate(zip(a, b)):
                if x != y: print("   line", i, "\n    hip", x, "\n    ref", y); break
            print("   lines hip", len(a), "ref", len(b), flush=True)
    for f in (rf,) + ((qf,) if qrec is not None else ()): os.remove(f)
print("paf fuzz done: %d iterations, %d failures" % (n_iter, bad))
sys.exit(1 if bad else 0)

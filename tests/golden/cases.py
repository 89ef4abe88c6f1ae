"""Seeded inputs shared by make_golden.py (which runs the REAL reference on them, in the build container)
and by the tests that replay them against the oracle (CPU) and the HIP path (GPU).  Everything is derived
from mmutil's counter-based generators, so the inputs regenerate bit-identically anywhere."""
import numpy as np

import mmutil as U

KNOWN_KMERS = [b"ACGTACGTACGTACGTACG", b"CGTACGTACGTACGTACGT", b"AAAAAAAAAAAAAAAAAAA", b"TTTTTTTTTTTTTTTTTTT",
               b"GATTACAGATTACAGATTA", b"TAATCTGTAATCTGTAATC", b"ACGTACGTACGTACGT"]


def hash_inputs():
    out = list(KNOWN_KMERS)
    for k in (11, 15, 16, 17, 19, 21, 24, 27, 31, 32):
        a = U.random_dna(900 + k, 40 * k)
        out += [bytes(a[i * k:(i + 1) * k]) for i in range(4)]
    return out


def sketch_cases():
    """(name, k, s, sequence)"""
    g = U.random_dna(101, 60000)
    return [
        ("random5000", 19, 130, g[:5000]),
        ("random5000_s498", 19, 498, g[5000:10000]),
        ("random10000_s40", 19, 40, g[10000:20000]),
        ("short300_s5", 19, 5, g[20000:20300]),
        ("tandem", 19, 130, U.tandem_repeat(7, 5000, 37)),
        ("n_runs", 19, 130, U.with_n_runs(g[30000:35000], 3, 6, 40)),
        ("lowercase", 19, 130, U.lowercase_some(g[35000:40000], 4)),
        ("k16", 16, 60, g[40000:41000]),
        ("k27", 27, 100, g[41000:46000]),
        ("fewer_than_s", 19, 498, g[46000:46300]),
        ("polyA", 19, 130, np.frombuffer(b"A" * 3000, dtype=np.uint8).copy()),
        ("allN", 19, 130, np.frombuffer(b"N" * 3000, dtype=np.uint8).copy()),
    ]


def minmer_cases():
    """(name, k, w, s, sequence)"""
    g = U.random_dna(202, 200000)
    rep = np.concatenate([U.mutate(g[:8000], 300 + i, 0.01) for i in range(6)])
    return [
        ("random_w5000", 19, 5000, 130, g[:60000]),
        ("random_w1000_s50", 19, 1000, 50, g[60000:90000]),
        ("repeat_w5000", 19, 5000, 130, rep),
        ("tandem_w500_s100", 19, 500, 100, U.tandem_repeat(9, 12000, 41)),
        ("n_runs_w1000", 19, 1000, 60, U.with_n_runs(g[90000:110000], 5, 8, 60)),
        ("early_N", 19, 300, 400, np.concatenate([g[110000:110005], np.frombuffer(b"N", dtype=np.uint8), g[110006:110400]])),
        ("shorter_than_w", 19, 5000, 130, g[120000:123000]),
        ("w10000_s20", 19, 10000, 20, g[130000:170000]),
    ]


def session_case():
    """a small reference + reads for the L1/L2 leg: (contigs, reads, params)"""
    cs = [U.random_dna(401, 160000), U.random_dna(402, 120000), U.random_dna(403, 3000)]
    blk = U.mutate(cs[0][50000:70000], 77, 0.03)
    cs[1][20000:20000 + len(blk)] = blk                        # a diverged duplicate -> two candidate regions
    contigs = [("chrA", cs[0]), ("chrB", cs[1]), ("tiny", cs[2])]
    reads = [(n, a) for n, a, _ in U.sample_reads([c for _, c in contigs], 5, 24, 10000, 0.10)]
    reads += [(n + "b", a) for n, a, _ in U.sample_reads([c for _, c in contigs], 6, 8, 7777, 0.04)]
    reads += [("dup", cs[0][52000:64000].copy()), ("unrelated", U.random_dna(404, 10000)),
              ("withN", U.with_n_runs(cs[1][70000:80000], 8, 4, 30))]
    params = dict(k=19, segLength=5000, sketchSize=130, pi=0.85, kmerPct=0.001)
    return contigs, reads, params


def fragments_of(read_len, L):
    """Map::mapModule's cut (computeMap.hpp:587-671): full segments + one overlapping tail"""
    if read_len <= L:
        return [(0, read_len)]
    fr = [(i * L, L) for i in range(read_len // L)]
    if read_len % L:
        fr.append((read_len - L, L))
    return fr


def paf_cases():
    """end-to-end command lines: (name, reference records, query records or None for self-map, extra argv)"""
    cs = [U.random_dna(601, 300000), U.random_dna(602, 200000), U.random_dna(603, 120000), U.random_dna(604, 3000)]
    blk = U.mutate(cs[0][40000:75000], 78, 0.03)
    cs[1][30000:30000 + len(blk)] = blk
    inv = U.revcomp(cs[0][150000:180000])
    cs[2][50000:50000 + len(inv)] = U.mutate(inv, 79, 0.02)[:30000]
    ref = [("chr1", cs[0]), ("chr2", cs[1]), ("chr3", cs[2]), ("tiny", cs[3])]
    reads = [(n, a) for n, a, _ in U.sample_reads(cs[:3], 15, 40, 10000, 0.10)]
    reads += [(n + "_s", a) for n, a, _ in U.sample_reads(cs[:3], 16, 12, 6200, 0.06)]
    reads += [(n + "_l", a) for n, a, _ in U.sample_reads(cs[:3], 17, 6, 31000, 0.04)]
    reads += [("short3k", cs[1][5000:8000].copy()), ("tiny12", cs[1][100:112].copy()), ("unrelated", U.random_dna(605, 12000)),
              ("chimera", np.concatenate([cs[0][10000:22000], U.revcomp(cs[2][20000:31000])]))]
    # haplotype-like set for the CI-style all-vs-all run (-Y '#')
    base = [U.random_dna(610 + c, 70000) for c in range(3)]
    hap = []
    for h in range(3):
        for c in range(3):
            a = base[c] if h == 0 else U.mutate(base[c], 700 + 10 * h + c, 0.015 * h)
            hap.append(("S%d#1#chr%d" % (h, c), a))
    # long noisy reads for the large-sketch command line (--dense at -s 20000: sketchSize 1998, parseCmdArgs.hpp:626-630)
    long_reads = [(n + "_L", a) for n, a, _ in U.sample_reads(cs[:3], 18, 10, 45000, 0.12)] + [(n + "_M", a) for n, a, _ in U.sample_reads(cs[:3], 19, 6, 20000, 0.15)]
    long_reads += [("unrelated_L", U.random_dna(606, 41000)), ("short19k", cs[0][200000:219000].copy())]
    asm = [("q_chr1", U.mutate(cs[0], 90, 0.01)), ("q_chr2", U.revcomp(U.mutate(cs[1], 91, 0.01))), ("q_chr3", U.mutate(cs[2], 92, 0.02))]
    return [
        ("default", ref, reads, []),
        ("pi90_n2", ref, reads, ["--pi", "90", "-n", "2"]),
        ("dense_pi80", ref, reads, ["--dense", "--pi", "80"]),
        ("nomerge", ref, reads, ["-M"]),
        ("filter_none", ref, reads, ["-f", "none"]),
        ("nohg_dropK", ref, reads, ["--noHgFilter", "-K"]),
        ("legacy_pct", ref, reads, ["--legacy"]),
        ("asm_one2one", ref, asm, ["--pi", "95", "-s", "10000", "-f", "one-to-one", "-J", "40"]),
        ("allvsall_Y", hap, None, ["--pi", "95", "-n", "1", "-Y", "#"]),
        ("allvsall_X_lower", hap, hap, ["--pi", "90", "-X", "--lowerTriangular", "-n", "3"]),
        ("dense_pi80_s20000", ref, long_reads, ["--dense", "--pi", "80", "-s", "20000"]),
        ("nosplit", ref, reads, ["--noSplit"]),                      # reads of 6.2, 10, 12 and 31 kbp at segLength 5000: windowLen != 0
        ("nosplit_pi90_n3", ref, reads + long_reads[:6], ["--noSplit", "--pi", "90", "-n", "3", "-s", "3000"]),
        # k-mers of more than 32 bases (the reference hashes any length, commonFunc.hpp:138): assembly-like queries at 1-2 % divergence
        ("k40_asm", ref, asm, ["-k", "40", "--pi", "95", "-s", "10000", "-f", "none"]),
        ("k57_asm_pi97", ref, asm, ["-k", "57", "--pi", "97", "-s", "5000", "-n", "2"]),
    ]


def paf_list_cases():
    """command lines whose inputs are LISTS of files (--rl / --ql, parseCmdArgs.hpp:293-313): (name, [records of reference file i], [records
    of query file j], extra argv).  configs[4] of BASELINE.json has this form: --dense --pi 80, 20 kbp reads at 15-20 % error, ten references."""
    cs = [U.random_dna(900 + i, n) for i, n in enumerate((90000, 70000, 110000, 60000, 80000, 100000, 65000, 75000, 95000, 85000))]
    blk = U.mutate(cs[0][20000:50000], 91, 0.05); cs[7][10000:10000 + len(blk)] = blk[:len(cs[7]) - 10000][:len(blk)]
    ref_files = [[("f%d_chr" % i, c)] for i, c in enumerate(cs)]
    ref_files[3].append(("f3_extra", U.random_dna(950, 30000)))
    reads = []
    for j, err in enumerate((0.15, 0.18, 0.20)):
        reads += [("n%d_%s" % (j, n), a) for n, a, _ in U.sample_reads(cs, 70 + j, 14, 20000, err)]
    q_files = [reads[:20], reads[20:]]
    return [("rl_dense_pi80_20kbp", ref_files, q_files, ["--dense", "--pi", "80"]),
            ("rl_default_n3", ref_files, q_files, ["-n", "3"])]

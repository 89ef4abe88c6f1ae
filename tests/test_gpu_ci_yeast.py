"""BASELINE configs[0] at its own scale: the reference's CI job (.github/workflows/build-and-test.yml:67-74) maps data/scerevisiae8.fa.gz --
8 yeast haplotypes x 17 sequences, 96 Mbp -- against itself with `--pi 95 -n 1 -Y '#'` and requires every sequence to be covered to 92 %
by the union of its query and target intervals (scripts/test.sh:7-36).  The FASTA is not in the reference tree, its .fai index is: the
names and lengths here are those of the index (tests/golden/scerevisiae8_lengths.tsv, made by make_ci_lengths.py), the sequences are
synthetic -- one random base genome, every haplotype a copy with 0.4 % substitutions (yeast strains differ by 0.5-1 %) cut into 17
sequences of ITS lengths -- written
gzip-compressed as the CI's input is.  The same command line through `mashmap_hip` and the stock binary (oracle/_ref/mashmap_ref, built
from the reference's sources): PAF bytes equal, and the coverage check of scripts/test.sh, restated in Python, passes on them."""
import gzip
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import mmutil as U

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_BIN = os.path.join(ROOT, "mashmap_amd", "lib", "mashmap_hip")
LENGTHS = os.path.join(ROOT, "tests", "golden", "scerevisiae8_lengths.tsv")


def ci_sequences():
    recs = [l.rstrip("\n").split("\t") for l in open(LENGTHS)]
    recs = [(n, int(l)) for n, l in recs]
    assert len(recs) == 136 and sum(l for _, l in recs) == 96255507
    haps = []
    for n, _ in recs:
        if n.split("#")[0] not in haps:
            haps.append(n.split("#")[0])
    assert len(haps) == 8 and all(sum(1 for n, _ in recs if n.split("#")[0] == h) == 17 for h in haps)
    # the strains' chromosomes differ in length by up to a third (translocations): a haplotype is the base genome -- as long as the
    # shortest haplotype, read cyclically -- cut into 17 pieces of ITS lengths, so that every base of every haplotype has a homolog in the others
    total = min(sum(l for n, l in recs if n.split("#")[0] == h) for h in haps)
    rng = np.random.default_rng(2026)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(0, 4, total)]
    out = []
    for h in haps:
        at = 0
        for n, l in recs:
            if n.split("#")[0] != h:
                continue
            a = np.take(genome, np.arange(at, at + l) % total)
            at += l
            hit = rng.random(l) < 0.004 * 4.0 / 3.0
            a[hit] = acgt[rng.integers(0, 4, int(hit.sum()))]
            out.append((n, a))
    return out


def coverage(paf_bytes, lengths):
    """scripts/test.sh: per sequence, the bases covered by the union of its intervals as a query (columns 3-4) and as a target (8-9)"""
    iv = {n: [] for n in lengths}
    for line in paf_bytes.decode().splitlines():
        f = line.split("\t")
        iv[f[0]].append((int(f[2]), int(f[3])))
        iv[f[5]].append((int(f[7]), int(f[8])))
    cov = {}
    for n, xs in iv.items():
        xs.sort()
        tot, end = 0, 0
        for a, b in xs:
            a, b = max(a, end, 0), min(b, lengths[n])
            if b > a:
                tot += b - a; end = b
        cov[n] = tot / lengths[n]
    return cov


def test_ci_yeast_self_map_at_its_own_scale(tmp_path):
    if not os.path.exists(U.REF_BIN):
        pytest.skip("oracle/_ref/mashmap_ref (the stock binary) is not here: it is built where /root/reference exists and shipped by gpurun")
    assert os.path.exists(HIP_BIN), "mashmap_hip not built"
    seqs = ci_sequences()
    fa = str(tmp_path / "scerevisiae8_synthetic.fa.gz")
    t0 = time.time()
    with gzip.open(fa, "wb", compresslevel=1) as f:
        for n, a in seqs:
            f.write(b">" + n.encode() + b"\n")
            full = len(a) // 80 * 80
            f.write(np.concatenate([a[:full].reshape(-1, 80), np.full((full // 80, 1), 10, dtype=np.uint8)], axis=1).tobytes())
            if len(a) > full:
                f.write(a[full:].tobytes() + b"\n")
    sys.path.insert(0, ROOT)
    import bench as B
    threads = str(max(4, min(32, 2 * B.usable_cpus())))
    args = ["-r", fa, "-q", fa, "--pi", "95", "-n", "1", "-Y", "#", "-t", threads]
    out = {}
    for name, exe in (("hip", HIP_BIN), ("stock", U.REF_BIN)):
        t1 = time.time()
        p = subprocess.run([exe] + args + ["-o", str(tmp_path / (name + ".paf"))], capture_output=True, text=True)
        assert p.returncode == 0, "%s %s\n%s" % (exe, " ".join(args), p.stderr[-2000:])
        tm = [l.split("] ")[-1] for l in p.stderr.splitlines() if "time spent" in l]
        out[name] = open(str(tmp_path / (name + ".paf")), "rb").read()
        print("\n[ci yeast] %s: %.1f s wall, %s, %d PAF lines" % (name, time.time() - t1, tm, out[name].count(b"\n")), flush=True)
    print("[ci yeast] 136 sequences, 96.3 Mbp, gzip FASTA written in %.0f s" % (t1 - t0))
    assert out["hip"] == out["stock"], "PAF differs from the stock binary's"
    lengths = {n: len(a) for n, a in seqs}
    cov = coverage(out["hip"], lengths)
    low = {n: round(c, 4) for n, c in cov.items() if c < 0.92}
    print("[ci yeast] coverage: min %.4f (%s), mean %.4f" % (min(cov.values()), min(cov, key=cov.get), sum(cov.values()) / len(cov)))
    assert not low, "scripts/test.sh would fail: low coverage for %r" % low

"""The CPU oracle against the REAL reference built from /root/reference (oracle/_ref/libmashmap_ref.so) on larger seeded
inputs than the committed golden vectors.  Skipped where oracle/_ref is not present (the `ref` fixture)."""
import os
import tempfile

import numpy as np
import pytest

import mmutil as U


def test_get_hash(oracle, ref):
    a = U.random_dna(1, 4000)
    for k in (8, 15, 16, 17, 19, 23, 24, 31, 32, 33, 40, 47, 48, 49, 57, 63, 64):
        for i in range(0, 60):
            s = bytes(a[i * 41:i * 41 + k])
            assert oracle.get_hash(s) == ref.get_hash(s)


@pytest.mark.parametrize("seed", range(6))
def test_sketch_sequence(oracle, ref, seed):
    n = [5000, 300, 2000, 10000, 777, 5000][seed]
    a = U.random_dna(10 + seed, n)
    variants = [a, U.with_n_runs(a, seed, 4, 25), U.lowercase_some(a, seed), U.tandem_repeat(seed, n, 29 + seed)]
    for v in variants:
        for k, s in ((19, 130), (19, 498), (16, 50), (21, 20)):
            assert oracle.sketch_sequence(v, k, s, seed) == ref.sketch_sequence(v, k, s, seed)


@pytest.mark.parametrize("k,s,n", [(33, 90, 3000), (40, 130, 5000), (57, 80, 4000), (64, 100, 4000), (19, 9000, 60000), (40, 8500, 50000)])
def test_sketch_sequence_long_kmers_and_large_sketches(oracle, ref, k, s, n):
    """the parameter ranges the HIP path covers with its wide-strip kernels (k 33..64) and its global-memory sketch kernel (s > 8 190)"""
    a = U.random_dna(300 + k, n)
    for v in (a, U.with_n_runs(a, 2, 4, 25)):
        assert oracle.sketch_sequence(v, k, s, 1) == ref.sketch_sequence(v, k, s, 1)


@pytest.mark.parametrize("cfg", [(5000, 130, 200000, "random"), (5000, 130, 120000, "repeat"), (1000, 50, 100000, "nruns"),
                                 (500, 100, 60000, "tandem"), (10000, 20, 150000, "random"), (5000, 498, 80000, "nruns")])
def test_add_minmers(oracle, ref, cfg):
    w, s, n, mode = cfg
    a = U.random_dna(77 + w + s, n)
    if mode == "repeat":
        a = np.concatenate([U.mutate(a[:n // 8], 5 + i, 0.01) for i in range(8)])
    elif mode == "nruns":
        a = U.with_n_runs(a, 3, 12, 80); a[4] = ord("N")
    elif mode == "tandem":
        a = U.tandem_repeat(5, n, 53)
    got, exp = oracle.add_minmers(a, 19, w, s, 2), ref.add_minmers(a, 19, w, s, 2)
    assert len(got) == len(exp) and len(exp) > 0
    for f in ("hash", "wpos", "wpos_end", "seqId", "strand"):
        assert np.array_equal(got[f], exp[f]), f


@pytest.mark.parametrize("k,w,s", [(33, 1000, 40), (40, 5000, 130), (48, 2000, 60), (49, 3000, 75), (64, 5000, 100)])
def test_add_minmers_long_kmers(oracle, ref, k, w, s):
    """the reference side of the index at k-mer sizes beyond 32 (the winnow kernels' wide strips are checked against this oracle)"""
    a = U.with_n_runs(U.random_dna(900 + k, 90000), 3, 8, 60)
    got, exp = oracle.add_minmers(a, k, w, s, 1), ref.add_minmers(a, k, w, s, 1)
    assert len(got) == len(exp) and len(exp) > 0
    for f in ("hash", "wpos", "wpos_end", "seqId", "strand"):
        assert np.array_equal(got[f], exp[f]), f


@pytest.mark.parametrize("mode", ["default", "dense", "prefix"])
def test_session_and_fragments(oracle, ref, mode):
    k, L, s, pi, flags, delim = 19, 5000, 130, 0.85, U.FLAG_HG, b"\0"
    if mode == "dense":
        s, pi = 498, 0.80
    cs = [U.random_dna(500 + i, n) for i, n in enumerate((250000, 180000, 90000))]
    blk = U.mutate(cs[0][30000:60000], 9, 0.04); cs[1][10000:10000 + len(blk)] = blk
    names = ["chr0", "chr1", "chr2"]
    if mode == "prefix":
        names = ["S1#1#a", "S1#1#b", "S2#1#a"]; flags |= U.FLAG_SKIP_PREFIX; delim = b"#"
    contigs = list(zip(names, cs))
    reads = [(nm, a) for nm, a, _ in U.sample_reads(cs, 3, 25, 10000, 0.12 if mode == "dense" else 0.08)]
    if mode == "prefix":
        reads = [("S2#1#q%d" % i, a) for i, (_, a) in enumerate(reads)]
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "r.fa")
        U.write_fasta(fa, contigs)
        hr = ref.session([fa], k, L, s, pi, U.FILTER_MAP, flags, delim, 0.001)
        ho = oracle.session(contigs, k, L, s, pi, U.FILTER_MAP, flags, delim, 0.001)
        io_, ir = oracle.index_array(ho), ref.index_array(hr)
        assert len(io_) == len(ir)
        for f in ("hash", "wpos", "wpos_end", "seqId", "strand"):
            assert np.array_equal(io_[f], ir[f])
        ko, co = oracle.keys(ho); kr, cr = ref.keys(hr)
        assert np.array_equal(ko, kr) and np.array_equal(co, cr)
        for key in kr[:: max(1, len(kr) // 200)]:
            assert oracle.lookup(ho, key) == ref.lookup(hr, key)
        assert oracle.cutoffs(ho) == ref.cutoffs(hr)
        nl2 = 0
        for ri, (nm, a) in enumerate(reads):
            for off in range(0, len(a) - L + 1, L):
                eo = oracle.map_fragment(ho, a[off:off + L], ri, nm.encode(), len(a), s)
                er = ref.map_fragment(hr, a[off:off + L], ri, nm.encode(), len(a), s)
                for key in ("sketch", "l1", "l2", "maps_i", "minimumHits", "sketchSize", "rawSketchSize"):
                    assert eo[key] == er[key], (mode, ri, off, key)
                # points: (seqId, pos, side) in order; which *hash* comes first among points that tie on all three is an
                # artefact of the reference's heap merge (computeMap.hpp:885-907) that nothing downstream reads
                assert [p[:3] for p in eo["points"]] == [p[:3] for p in er["points"]], (mode, ri, off)
                assert sorted(eo["points"]) == sorted(er["points"]), (mode, ri, off)
                for x, y in zip(eo["maps_f"], er["maps_f"]):
                    assert all(abs(p - q) <= 1e-6 for p, q in zip(x, y))
                nl2 += len(er["l2"])
        assert nl2 > 20
        ref.free(hr); oracle.free(ho)


@pytest.mark.parametrize("mode", ["default", "dup"])
def test_fragments_longer_than_the_segment_window_len(oracle, ref, mode):
    """--noSplit: a read longer than segLength is ONE fragment with windowLen = Q.len - segLength != 0 (computeMap.hpp:933, :1309): the
    hash_to_freq branches of computeL1CandidateRegions and computeL2MappedRegions, the heap of open records, the shifted coordinates.
    `dup`: a reference with exact repeats, so that hashes are open more than once inside the window (the counts matter)."""
    k, L, s, pi = 19, 5000, 130, 0.85
    cs = [U.random_dna(700 + i, n) for i, n in enumerate((300000, 200000))]
    if mode == "dup":
        unit = cs[0][40000:52000].copy()
        for j in range(3):
            m = U.mutate(unit, 90 + j, 0.002 * j)[:12000]
            cs[0][100000 + j * 20000:100000 + j * 20000 + len(m)] = m
        cs[1][50000:62000] = unit
    contigs = list(zip(["chr0", "chr1"], cs))
    reads = [(nm, a) for nm, a, _ in U.sample_reads(cs, 13, 10, 12345, 0.08)] + [(nm, a) for nm, a, _ in U.sample_reads(cs, 14, 6, 31000, 0.05)]
    reads += [("exact", cs[0][38000:61000].copy()), ("short", cs[1][1000:4000].copy()), ("just_over", cs[1][70000:75001].copy())]
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "r.fa")
        U.write_fasta(fa, contigs)
        fl = U.FLAG_HG | U.FLAG_NOSPLIT
        hr = ref.session([fa], k, L, s, pi, U.FILTER_MAP, fl, b"\0", 0.001)
        ho = oracle.session(contigs, k, L, s, pi, U.FILTER_MAP, fl, b"\0", 0.001)
        nl2 = 0
        for ri, (nm, a) in enumerate(reads):
            eo = oracle.map_fragment(ho, a, ri, nm.encode(), len(a), s)
            er = ref.map_fragment(hr, a, ri, nm.encode(), len(a), s)
            for key in ("sketch", "l1", "l2", "maps_i", "minimumHits", "sketchSize", "rawSketchSize"):
                assert eo[key] == er[key], (mode, ri, nm, key, eo[key][:3] if isinstance(eo[key], list) else eo[key], er[key][:3] if isinstance(er[key], list) else er[key])
            nl2 += len(er["l2"])
        assert nl2 >= 10
        ref.free(hr); oracle.free(ho)

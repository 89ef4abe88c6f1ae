"""a5-a7 through the C ABI (mm_index_build) vs the CPU oracle: minmerIndex (addMinmers, commonFunc.hpp:302),
the lookup map (Sketch::index, winSketch.hpp:379) and the frequent-seed set (winSketch.hpp:410-504).
Bit-exact, including record order."""
import numpy as np
import pytest

import mmutil as U

pytestmark = pytest.mark.gpu


def _compare(oracle, contigs, k=19, L=5000, s=130, kmerPct=0.001):
    from mashmap_amd import capi
    h = oracle.session(contigs, k, L, s, 0.85, U.FILTER_MAP, U.FLAG_HG, b"\0", kmerPct)
    e = oracle.export_index(h)
    ctx = capi.Context(k=k, segLength=L, sketchSize=s)
    ctx.index_build([a for _, a in contigs], kmerPct=kmerPct)
    g = ctx.index_download()
    assert len(g["minmers"]) == len(e["minmers"])
    for fld in ("hash", "wpos", "wpos_end", "seqId", "strand"):
        assert (g["minmers"][fld] == e["minmers"][fld]).all(), fld
    assert (g["keys"] == e["keys"]).all() and (g["offsets"] == e["offsets"]).all()
    for fld in ("pos", "hash", "seqId", "side"):
        assert (g["points"][fld] == e["points"][fld]).all(), fld
    assert sorted(g["freq"].tolist()) == sorted(e["freq"].tolist())
    assert g["freqThreshold"] == oracle.f("session_freq_threshold")(h)
    ctx.close(); oracle.free(h)
    return len(e["minmers"]), len(e["freq"])


def test_index_random_contigs(oracle):
    contigs = [("c%d" % i, U.random_dna(300 + i, n)) for i, n in enumerate([250000, 4000, 120000, 10, 5000, 5018])]
    n, _ = _compare(oracle, contigs)
    assert n > 15000


def test_index_adversarial(oracle):
    a = U.tandem_repeat(1, 150000, 1777); a[::997] = ord("A")
    b = U.with_n_runs(U.random_dna(2, 120000), 3, 20, 60); b[2] = ord("N"); b[17] = ord("n")
    b[60000:63000] = np.tile(np.frombuffer(b"AC", dtype=np.uint8), 1500)
    c = U.lowercase_some(U.with_n_runs(U.random_dna(3, 90000), 9, 10, 25), 4)
    d = U.tandem_repeat(5, 40000, 47)          # period divides w-k+1 = 4982: departure and arrival of one hash in the same step
    _compare(oracle, [("a", a), ("b", b), ("c", c), ("d", d)])


def test_index_frequent_seeds(oracle):
    unit = U.random_dna(81, 20000)
    rep = np.concatenate([U.mutate(unit, 200 + i, 0.01) for i in range(12)])
    n, nf = _compare(oracle, [("rep", rep), ("uniq", U.random_dna(82, 100000))], kmerPct=0.5)
    assert nf > 0


@pytest.mark.parametrize("k,s,L", [(16, 50, 1000), (19, 498, 5000), (19, 20, 10000), (21, 100, 500), (40, 60, 1000), (64, 100, 2000)])
def test_index_parameter_grid(oracle, k, s, L):
    contigs = [("x", U.random_dna(7 + k, 150000)), ("y", U.tandem_repeat(8, 60000, 311)), ("z", U.with_n_runs(U.random_dna(9, 50000), 2, 6, 40))]
    _compare(oracle, contigs, k, L, s)


@pytest.mark.parametrize("k,s,L", [(19, 130, 5000), (19, 498, 5000), (16, 50, 1000), (19, 2000, 20000), (21, 100, 500)])
def test_index_with_the_blocked_window_sketch_at_small_sizes(oracle, monkeypatch, k, s, L):
    """the HBM form of the index build's window sketch (k_winnow_tiles<.., GSK>: blocks of 64 entries under an LDS directory, used beyond
    sketchSize 4 096) forced at ordinary sizes, where its blocks split, empty and merge thousands of times per tile: the same records as
    the oracle's addMinmers, on random, tandem-repeat and N-rich contigs"""
    monkeypatch.setenv("MM_WINNOW_GSK", "1")
    contigs = [("x", U.random_dna(70 + k, 150000)), ("y", U.tandem_repeat(80, 60000, 311)), ("z", U.with_n_runs(U.random_dna(90, 50000), 2, 6, 40))]
    _compare(oracle, contigs, k, L, s)


def test_index_at_the_largest_lds_sketch_size_and_beyond(oracle):
    """up to sketchSize 4 096 the device index build keeps a window's sketch as a sorted array in LDS, beyond that as blocks in HBM
    (k_winnow_tiles<.., GSK>; the reference's --dense derives 9 998 at 100 kbp segments and 19 998 at 200 kbp, parseCmdArgs.hpp:626-630,
    and takes any size).  Device-built index against the oracle, record for record, at 4 096 / 4 097 (last LDS, first HBM size) and 19 998
    (more than LDS could hold at all); 65 536 -- more seeds than the literal kernels' 16-bit seed numbers -- is refused by mm_create."""
    from mashmap_amd import capi
    contigs = [("c0", U.random_dna(901, 330000)), ("c1", U.with_n_runs(U.random_dna(902, 250000), 2, 20, 300))]
    for s in (4096, 4097):
        nm, _ = _compare(oracle, contigs, k=19, L=100000, s=s)
        assert nm > s
    contigs = [("c0", U.random_dna(903, 430000)), ("c1", U.with_n_runs(U.random_dna(904, 330000), 2, 20, 300))]
    nm, _ = _compare(oracle, contigs, k=19, L=200000, s=19998)
    assert nm > 19998
    with pytest.raises(capi.MashmapError, match="sketchSize 65536 is beyond 65535"):
        capi.Context(k=19, segLength=400000, sketchSize=65536)

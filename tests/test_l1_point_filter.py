"""The interval-point pre-filter of the HBM point path (tests/l1filter.py states the rule; mm_map.hip k_filter_points is the device form) must
leave computeL1CandidateRegions' output (computeMap.hpp:916-1116) exactly as it is: checked here on the CPU against the oracle's literal L1
over fuzzed point sets -- clusters that reach minimumHits, scattered noise, long merged intervals, contigs whose last and first points share a
position (the reference groups by `pos` alone), with and without the HG filter.  Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

import l1filter
import mmutil as U

L1_DT = np.dtype([("seqId", "<i4"), ("rangeStartPos", "<i4"), ("rangeEndPos", "<i4"), ("intersectionSize", "<i4")])


def _points(seq, o, c):
    pts = np.zeros(2 * len(seq), dtype=U.POINT_DT)
    pts["seqId"][0::2] = seq; pts["pos"][0::2] = o; pts["side"][0::2] = 1
    pts["seqId"][1::2] = seq; pts["pos"][1::2] = c; pts["side"][1::2] = -1
    order = np.lexsort((pts["side"], pts["pos"], pts["seqId"]))
    return np.ascontiguousarray(pts[order])


def _l1(orc, h, seq, o, c, qs, min_hits):
    pts = _points(seq, o, c)
    out = np.zeros(4096, dtype=L1_DT)
    fn = orc.lib.orc_session_l1_from_points
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = fn(h, pts.ctypes.data, len(pts), qs, 5000, min_hits, out.ctypes.data, len(out))
    assert 0 <= n <= len(out)
    return out[:n].tobytes(), n


def _scenario(rng, kind):
    ncontig = int(rng.integers(1, 6))
    clen = int(rng.choice([9000, 40000, 400000]))
    seq, o, c = [], [], []

    def add(q, a, ln):
        seq.append(q); o.append(int(a)); c.append(int(a + ln))
    for _ in range(int(rng.integers(0, 5))):                       # loci where many intervals overlap
        q = int(rng.integers(0, ncontig)); at = int(rng.integers(0, clen))
        for _ in range(int(rng.integers(2, 40))):
            add(q, at + int(rng.integers(-3000, 3000)) if at > 3000 else at + int(rng.integers(0, 3000)), int(rng.integers(1, 5000)))
    for _ in range(int(rng.integers(0, 300 if kind != "sparse" else 30))):   # scattered single hits
        add(int(rng.integers(0, ncontig)), int(rng.integers(0, clen)), int(rng.integers(1, 5000)))
    if kind == "long":                                             # merged windows of one hash: several segment lengths
        for _ in range(int(rng.integers(1, 4))):
            add(int(rng.integers(0, ncontig)), int(rng.integers(0, clen)), int(rng.integers(5000, 60000)))
    if kind == "seam":                                             # last point of a contig at the position of the next one's first
        p = int(rng.integers(100, 5000))
        for q in range(ncontig):
            add(q, p, int(rng.integers(1, 3000)))                  # many contigs open at p ...
            add(q, max(0, p - int(rng.integers(1, 3000))), 0 + int(rng.integers(1, 50)))
        m = max(c) + int(rng.integers(0, 3))
        for q in range(ncontig):
            add(q, m - int(rng.integers(1, 2000)), 0)              # ... and close at one position m (length fixed below)
            c[-1] = m
            for _ in range(int(rng.integers(0, 4))):
                add(q, p, int(rng.integers(1, 4000)))
    if kind == "seam2":
        # the case the boundary rule exists for: contig A's last point is the CLOSE of a lone interval at p, contig A + 1's first points are
        # a cluster opening at p -- the reference's sweep groups them together and reports the candidate under contig A; a filter that
        # drops the lone interval (it reaches no count) would move it to contig A + 1
        seq, o, c = [], [], []
        p = int(rng.integers(200, 9000)); A = int(rng.integers(0, 3)); k = int(rng.integers(2, 10))
        for _ in range(int(rng.integers(0, 10))):
            a = int(rng.integers(0, max(1, p - 150))); add(A, a, 1); c[-1] = min(p - 1, a + int(rng.integers(1, 120)))
        add(A, p - int(rng.integers(1, 150)), 1); c[-1] = p
        for _ in range(k):
            add(A + 1, p, int(rng.integers(50, 4000)))
        for _ in range(int(rng.integers(0, 20))):
            add(A + 1, p + int(rng.integers(1, 20000)), int(rng.integers(1, 4000)))
    seq = np.array(seq, dtype=np.int64); o = np.maximum(0, np.array(o, dtype=np.int64)); c = np.array(c, dtype=np.int64)
    c = np.maximum(c, o + 1)
    return seq, o, c


@pytest.mark.parametrize("flags", [U.FLAG_HG, 0])
def test_filtered_points_give_the_same_l1_candidates(oracle, flags):
    h = oracle.session([("c", U.random_dna(5, 30000))], 19, 5000, 60, 0.85, U.FILTER_MAP, flags)
    rng = np.random.default_rng(20260927 + flags)
    dropped = kept = with_candidates = 0
    for it in range(1500):
        kind = ("dense", "sparse", "long", "seam", "seam2")[it % 5]
        seq, o, c = _scenario(rng, kind)
        if len(seq) == 0:
            continue
        qs = int(rng.integers(10, 61))
        min_hits = int(rng.integers(1, 7))
        want, n = _l1(oracle, h, seq, o, c, qs, min_hits)
        with_candidates += n > 0
        for slots in (None, 64, 2048):
            m = l1filter.keep_mask(seq, o, c, min_hits, slots)
            got, _ = _l1(oracle, h, seq[m], o[m], c[m], qs, min_hits)
            assert got == want, (it, kind, min_hits, slots, int(m.sum()), len(m))
        m = l1filter.keep_mask(seq, o, c, min_hits, None)
        dropped += int((~m).sum()); kept += int(m.sum())
    oracle.free(h)
    assert with_candidates > 300 and dropped > 20000 and kept > 20000, (with_candidates, dropped, kept)


def test_the_bin_rule_alone_is_not_enough_at_a_contig_seam(oracle):
    """what the boundary rule is for (and that this test file can fail): without it the seam2 scenarios change their candidates"""
    h = oracle.session([("c", U.random_dna(5, 30000))], 19, 5000, 60, 0.85, U.FILTER_MAP, U.FLAG_HG)
    rng = np.random.default_rng(7)
    differ = 0
    for _ in range(200):
        seq, o, c = _scenario(rng, "seam2")
        min_hits = int(rng.integers(2, 6))
        want, _ = _l1(oracle, h, seq, o, c, 60, min_hits)
        m = l1filter.keep_mask(seq, o, c, min_hits, None)
        assert _l1(oracle, h, seq[m], o[m], c[m], 60, min_hits)[0] == want
        cnt = {}
        for i in range(len(seq)):
            for b in range(int(o[i]) >> l1filter.BIN_SHIFT, ((int(c[i]) - 1) >> l1filter.BIN_SHIFT) + 1):
                cnt[(int(seq[i]), b)] = cnt.get((int(seq[i]), b), 0) + 1
        pure = np.array([any(cnt[(int(seq[i]), b)] >= min_hits for b in range(int(o[i]) >> l1filter.BIN_SHIFT, ((int(c[i]) - 1) >> l1filter.BIN_SHIFT) + 1)) for i in range(len(seq))])
        differ += _l1(oracle, h, seq[pure], o[pure], c[pure], 60, min_hits)[0] != want
    oracle.free(h)
    assert differ > 20, differ

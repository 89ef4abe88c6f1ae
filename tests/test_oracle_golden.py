"""The CPU oracle (oracle/liboracle.so) against the golden vectors in tests/golden/, which were produced by the REAL
reference (tests/golden/make_golden.py, run where /root/reference exists).  This is what pins the oracle on machines
that do not have the reference.  Integers bit-exact; floats must match to the last bit as well (same glibc pow/log)
but are compared with the 1e-6 tolerance north_star states."""
import json
import os

import numpy as np
import pytest

import mmutil as U
from golden import cases as CS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLD, "golden.json")))


def test_hashes(oracle, gold):
    exp = {s: int(h) for s, h in gold["hashes"]}
    assert exp["ACGTACGTACGTACGTACG"] == 2819345507021956028
    for s in CS.hash_inputs():
        assert oracle.get_hash(s) == exp[s.decode()], s


def test_sketch_sequence(oracle, gold):
    for name, k, s, seq in CS.sketch_cases():
        exp = [(int(h), a, b, c, d) for h, a, b, c, d in gold["sketch"][name]]
        assert oracle.sketch_sequence(seq, k, s, 7) == exp, name
    assert len(gold["sketch"]["allN"]) == 0 and len(gold["sketch"]["polyA"]) == 1


def test_add_minmers(oracle):
    z = np.load(os.path.join(GOLD, "minmers.npz"))
    for name, k, w, s, seq in CS.minmer_cases():
        got = oracle.add_minmers(seq, k, w, s, 3)
        exp = z[name]
        assert len(got) == len(exp), name
        for f in ("hash", "wpos", "wpos_end", "seqId", "strand"):
            assert np.array_equal(got[f], exp[f]), (name, f)
    assert len(z["shorter_than_w"]) == 0


def test_stats(oracle, gold):
    st = gold["stats"]
    for i, v in enumerate(st["j2md_130"]):
        assert abs(oracle.f("j2md")(i / 130.0, 19) - float(v)) <= 1e-6
    for d, v in enumerate(st["md2j"]):
        assert abs(oracle.f("md2j")(d / 100.0, 19) - float(v)) <= 1e-6
    it = iter(st["md_lower_bound"])
    for d in (1, 5, 10, 15, 20):
        for s in (20, 130, 498):
            assert abs(oracle.f("md_lower_bound")(d / 100.0, s, 19, 0.95) - float(next(it))) <= 1e-6
    for key, tab in st["min_hits_relaxed"].items():
        s, pi = key.split("_")
        assert [oracle.f("min_hits_relaxed")(q, 19, int(pi) / 100.0) for q in range(1, int(s) + 1)] == tab
    for key, v in st["recommended_sketch_size"].items():
        pi, L, rs = key.split("_")
        assert oracle.f("recommended_sketch_size")(19, int(pi) / 100.0, int(L), int(rs)) == v, key


def test_session_index_and_fragments(oracle, gold):
    contigs, reads, P = CS.session_case()
    h = oracle.session(contigs, P["k"], P["segLength"], P["sketchSize"], P["pi"], U.FILTER_MAP, U.FLAG_HG, b"\0", P["kmerPct"])
    z = np.load(os.path.join(GOLD, "session_index.npz"))
    idx = oracle.index_array(h)
    assert len(idx) == gold["session"]["n_minmers"]
    for f in ("hash", "wpos", "wpos_end", "seqId", "strand"):
        assert np.array_equal(idx[f], z["minmers"][f]), f
    keys, counts = oracle.keys(h)
    assert np.array_equal(keys, z["keys"]) and np.array_equal(counts, z["counts"])
    assert oracle.f("session_freq_threshold")(h) == gold["session"]["freq_threshold"]
    assert oracle.cutoffs(h) == gold["session"]["cutoffs"]
    n = 0
    for fr in gold["session"]["fragments"]:
        name, a = reads[fr["read"]]
        e = oracle.map_fragment(h, a[fr["off"]:fr["off"] + fr["len"]], fr["read"], name.encode(), len(a), P["sketchSize"])
        assert e["sketchSize"] == fr["sketchSize"] and e["rawSketchSize"] == fr["rawSketchSize"]
        assert e["minimumHits"] == fr["minimumHits"]
        assert [[str(x[0]), x[4]] for x in e["sketch"][:6]] == fr["sketch_head"]
        assert [list(p[:3]) for p in e["points"]] == fr["points"]
        assert [list(x) for x in e["l1"]] == fr["l1"]
        assert [list(x) for x in e["l2"]] == fr["l2"]
        assert [list(x) for x in e["maps_i"]] == fr["maps_i"]
        for got, exp in zip(e["maps_f"], fr["maps_f"]):
            for g, x in zip(got, exp):
                assert abs(g - float(x)) <= 1e-6
        assert abs(e["kmerComplexity"] - float(fr["kmerComplexity"])) <= 1e-6
        n += len(fr["l2"])
    assert n > 60
    oracle.free(h)

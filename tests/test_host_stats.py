"""Host-side statistics of the product (mashmap_amd/host/mm_stats.hpp via the C ABI) vs the CPU oracle and,
when built, the real reference: j2md/md2j/md_lower_bound (map_stats.hpp:45-112) to 1e-6 (they are in fact
bit-identical), integer tables (minimum hits :144, sketch cut-offs computeMap.hpp:178, sketch size :234) exactly.
Runs without a GPU."""
import numpy as np
import pytest

from mashmap_amd import capi


@pytest.fixture(scope="module")
def lib():
    return capi.load()


def test_float_stats_match_oracle(lib, oracle):
    for k in (15, 16, 19, 21):
        for s in (20, 40, 130, 310, 498):
            for i in range(0, s + 1, max(1, s // 37)):
                j = np.float32(i / s)
                a, b = lib.mm_stat_j2md(j, k), oracle.f("j2md")(j, k)
                assert a == b and abs(a - b) <= 1e-6
                d = np.float32(a)
                assert lib.mm_stat_md2j(d, k) == oracle.f("md2j")(d, k)
                assert lib.mm_stat_md_lower_bound(d, s, k, 0.95) == oracle.f("md_lower_bound")(d, s, k, 0.95)


def test_known_answers_from_reference(lib):
    # SURVEY App. B.2 (values printed by the reference build): j2md(i/130, 19), md2j(0.15, 19)
    exp = {1: 0.197567791, 2: 0.168086424, 5: 0.128015533, 13: 0.0858161524, 65: 0.0211141743, 129: 0.000203583637}
    for i, v in exp.items():
        assert abs(lib.mm_stat_j2md(np.float32(i / 130.0), 19) - v) < 1e-8
    assert abs(lib.mm_stat_md2j(np.float32(0.15), 19) - 0.0233316924) < 1e-9
    assert lib.mm_stat_min_hits_relaxed(130, 19, 0.85) == 2


@pytest.mark.parametrize("k,pi", [(19, 0.85), (19, 0.80), (19, 0.95), (16, 0.90)])
def test_min_hits_table_matches_oracle(lib, oracle, k, pi):
    for s in list(range(1, 140)) + [220, 310, 498, 777, 1000]:
        assert lib.mm_stat_min_hits_relaxed(s, k, pi) == oracle.f("min_hits_relaxed")(s, k, pi), (s, k, pi)


@pytest.mark.parametrize("s", [20, 40, 130, 310, 498])
def test_sketch_cutoffs_match_oracle(oracle, s):
    import mmutil as U
    h = oracle.session([("c", U.random_dna(1, 30000))], 19, 5000, s, 0.85, U.FILTER_MAP, U.FLAG_HG)
    exp = oracle.cutoffs(h)
    oracle.free(h)
    assert capi.stat_sketch_cutoffs(s, 19).tolist() == exp


def test_recommended_sketch_size_matches_oracle_and_survey(lib, oracle):
    cases = [(19, 0.85, 5000, 100_000_000), (19, 0.95, 10000, 3_000_000_000), (19, 0.85, 5000, 3_000_000_000),
             (19, 0.85, 5000, 18446744072414584320), (16, 0.90, 2000, 5_000_000)]
    for k, pi, L, R in cases:
        assert lib.mm_stat_recommended_sketch_size(k, pi, L, R) == oracle.f("recommended_sketch_size")(k, pi, L, R)
    assert lib.mm_stat_recommended_sketch_size(19, 0.85, 5000, 100_000_000) == 130


def test_stats_match_real_reference(lib, ref):
    for s in (40, 130, 498):
        for i in range(0, s + 1, 7):
            j = np.float32(i / s)
            assert lib.mm_stat_j2md(j, 19) == ref.f("j2md")(j, 19)
        for q in range(1, s + 1, 3):
            assert lib.mm_stat_min_hits_relaxed(q, 19, 0.85) == ref.f("min_hits_relaxed")(q, 19, 0.85)


# ---- the GSL boundary, pinned by a third derivation (tests/gslcheck.py: scipy's CDFs, the reference's float mixing restated in numpy)
def test_gsl_boundary_minimum_hits_sketch_cutoffs_and_their_margins(lib):
    """estimateMinimumHitsRelaxed (map_stats.hpp:81-169) for every Q.sketchSize and Map::sketchCutoffs (computeMap.hpp:178-258) at every
    (k, pi, sketchSize) of BASELINE.json: equal to mm_stat_*, and no compared CDF value closer to its threshold than 1e-5 relative --
    seven orders of magnitude more than any of the three CDF derivations (product, GSL stand-in, scipy) can be off."""
    import gslcheck as G
    M = G.Margins()
    for k, pi, s in G.TABLE_CASES:
        step = 1 if s <= 220 else 3                 # the two largest tables: every third Q.sketchSize (the whole table: python tests/gslcheck.py)
        for q in list(range(1, s + 1, step)) + [s]:
            assert G.estimate_minimum_hits_relaxed(q, k, pi, G.CI, M) == lib.mm_stat_min_hits_relaxed(q, k, pi), (k, pi, q)
        assert G.sketch_cutoffs(s, k, M) == capi.stat_sketch_cutoffs(s, k).tolist(), (k, s)
    assert set(M.best) == {"md_lower_bound: binomial_Q < q2", "sketchCutoffs: prAboveCutoff > min_p"}
    assert M.floor() > 1e-5, M.best
    # the closest comparison of each kind once more at 60 digits: same side of the threshold
    for kind, (m, v, t, w) in M.best.items():
        again = G.confirm_with_mpmath(kind, w)
        assert (again > t) == (v > t) and abs(again - v) <= 1e-9 * t, (kind, v, again, t)


def test_gsl_boundary_recommended_sketch_size(lib):
    """recommendedSketchSize (map_stats.hpp:181-262) for SURVEY App. C's rows: the p-value crosses 1e-3 by at least a factor 5 at the
    sketch size that is chosen and at the one before it"""
    import gslcheck as G
    for k, pi, L, R, exp in G.SKETCH_SIZE_ROWS:
        M = G.Margins()
        assert G.recommended_sketch_size(k, pi, L, R, M) == exp == lib.mm_stat_recommended_sketch_size(k, pi, L, R)
        assert M.best["recommendedSketchSize: pValue <= 1e-3"][0] > 0.5, M.best

"""An independent re-derivation of the three integer results the reference computes through GNU GSL -- test infrastructure.

The reference calls gsl_cdf_binomial_Q (map_stats.hpp:98, :213), gsl_ran_hypergeometric_pdf (computeMap.hpp:194) and
gsl_cdf_hypergeometric_P (computeMap.hpp:213); GSL is in neither this image nor the reference tree, so the product
(mashmap_amd/host/mm_stats.hpp: log-space sums), the oracle and oracle/_ref (gsl_shim: continued fraction) all stand on
re-derivations.  The CDF values feed only comparisons against fixed thresholds (`< q2`, `<= 1e-3`, `> min_p`), so any
implementation accurate to far less than the distance of the nearest CDF value from its threshold yields the same integers.
This module restates the three routines in Python with the reference's float / double mixing (numpy float32 where the
reference has `float`), takes the CDFs from scipy.stats (Boost's incomplete beta / its own hypergeometric: a third
derivation), records how close every compared value comes to its threshold, and can confirm the closest ones with mpmath at
60 digits.  tests/test_host_stats.py asserts equality with mm_stat_* and a floor on the margins;

    python tests/gslcheck.py > profiles/r13_gsl_boundary_margins.txt

prints them."""
import math

import numpy as np
from scipy import stats

f32 = np.float32
CI = f32(0.95)                      # skch::fixed::confidence_interval (map_parameters.hpp:91)


def j2md(j, k):                     # map_stats.hpp:45
    j = f32(j)
    if j == 0:
        return f32(1.0)
    if j == 1:
        return f32(0.0)
    x = f32(f32(2) * j) / f32(f32(1) + j)             # float arithmetic
    return f32(1.0 - math.pow(float(x), 1.0 / k))     # std::pow(float, double) is the double pow


def md2j(d, k):                     # map_stats.hpp:63
    sim = f32(f32(1) - f32(d))
    p = math.pow(float(sim), float(k))
    return f32(p / (2.0 - p))


class Margins:
    """smallest relative distance |value - threshold| / threshold seen per kind of comparison, and where"""

    def __init__(self):
        self.best = {}

    def see(self, kind, value, threshold, where):
        m = abs(value - threshold) / threshold
        if kind not in self.best or m < self.best[kind][0]:
            self.best[kind] = (m, value, threshold, where)

    def floor(self):
        return min(v[0] for v in self.best.values()) if self.best else float("inf")


def binom_q(x, p, n):               # gsl_cdf_binomial_Q(x, p, n) = P[X > x]
    return float(stats.binom.sf(x, n, p))


def md_lower_bound(d, s, k, ci, M=None):             # map_stats.hpp:81-112, GSL branch
    q2 = f32((1.0 - float(ci)) / 2)
    p = md2j(d, k)
    x = max(int(math.ceil(f32(f32(s) * p))), 1)      # int * float is a float product
    while x <= s:
        cdf_complement = binom_q(x - 1, float(p), s)
        if M is not None:
            M.see("md_lower_bound: binomial_Q < q2", cdf_complement, float(q2), dict(s=s, k=k, x=x, p=float(p)))
        if cdf_complement < float(q2):
            x -= 1
            break
        x += 1
    return j2md(f32(f32(x) / f32(s)), k)


def estimate_minimum_hits(s, k, pi):                 # map_stats.hpp:122
    mash_dist = f32(1.0 - float(f32(pi)))
    return int(math.ceil(1.0 * s * float(md2j(mash_dist, k))))


def estimate_minimum_hits_relaxed(s, k, pi, ci=CI, M=None):     # map_stats.hpp:144
    first = estimate_minimum_hits(s, k, pi)
    relaxed = first
    for i in range(first, -1, -1):
        d = j2md(f32(1.0 * i / s), k)
        id_upper = f32(1.0 - float(md_lower_bound(d, s, k, ci, M)))
        if id_upper >= f32(pi):
            relaxed = i
        else:
            break
    return relaxed


def estimate_pvalue(s, k, alphabet, pi, length_query, length_reference, ci=CI, M=None):     # map_stats.hpp:181
    kmer_space = math.pow(alphabet, k)
    px = 1.0 / (1.0 + kmer_space / length_query)
    r = px * px / (px + px - px * px)
    x = estimate_minimum_hits_relaxed(s, k, pi, ci, M)
    return float(length_reference) * (1.0 if x == 0 else binom_q(x - 1, r, s))


def recommended_sketch_size(k, pi, segment_length, length_reference, M=None):                # map_stats.hpp:234 with the fixed:: arguments
    length_query = segment_length - k
    s = 10
    while s < length_query:
        pv = estimate_pvalue(s, k, 4, pi, length_query, length_reference, CI, M)
        if M is not None:
            M.see("recommendedSketchSize: pValue <= 1e-3", pv, 1e-3, dict(s=s, k=k, pi=pi, L=segment_length, R=length_reference))
        if pv <= 1e-3:
            break
        s += 10
    return s


def sketch_cutoffs(sketch_size, k, M=None, ani_diff_conf=f32(0.999)):                        # computeMap.hpp:178-258, deltaANI == 0 (the default)
    ss = int(min(float(sketch_size), 1000.0))
    min_p = float(f32(f32(1) - ani_diff_conf))
    y = np.arange(ss + 1)
    # P[c][y]  = gsl_ran_hypergeometric_pdf(y, ss, ss - c, c): y marked among c drawn from ss marked + (ss - c) unmarked
    # S[c][y]  = 1 - gsl_cdf_hypergeometric_P(y - 1, ss, ss - c, c) = P[Y >= y]  (1 for y = 0; P is 1 once y - 1 >= c)
    P = np.zeros((ss + 1, ss + 1))
    S = np.zeros((ss + 1, ss + 1))
    for c in range(ss + 1):
        h = stats.hypergeom(2 * ss - c, ss, c)
        P[c, :c + 1] = h.pmf(y[:c + 1])
        S[c, :c + 1] = 1.0 - np.where(y[:c + 1] >= 1, h.cdf(y[:c + 1] - 1), 0.0)
    pr = P @ S.T                    # pr[cmax][ci] = sum over ymax of P[cmax][ymax] * S[ci][ymax]; the early return only shortens a monotone sum
    cut = [1] * (ss + 1)
    for cmax in range(1, ss + 1):
        ok = pr[cmax, :ss] > min_p
        lo = int(np.argmax(ok)) if ok.any() else ss
        cut[cmax] = 1 if lo == 0 else lo
    if M is not None:               # every (cmax, ci) pair, not just the ones next to a cut-off: the binary search may visit any of them
        d = np.abs(pr[1:, :ss] - min_p)
        cmax, ci = np.unravel_index(int(np.argmin(d)), d.shape)
        M.see("sketchCutoffs: prAboveCutoff > min_p", float(pr[cmax + 1, ci]), min_p, dict(ss=ss, cmax=int(cmax) + 1, ci=int(ci)))
    return cut


def confirm_with_mpmath(kind, where):
    """the value of the closest comparison of a kind again, at 60 digits from exact binomial coefficients"""
    import mpmath as mp
    mp.mp.dps = 60
    if kind.startswith("sketchCutoffs"):
        ss, cmax, ci = where["ss"], where["cmax"], where["ci"]

        def pdf(yv, c):
            return mp.binomial(ss, yv) * mp.binomial(ss - c, c - yv) / mp.binomial(2 * ss - c, c) if 0 <= yv <= c else mp.mpf(0)
        tot = mp.mpf(0)
        for ymax in range(cmax + 1):
            tot += pdf(ymax, cmax) * (1 - sum(pdf(i, ci) for i in range(ymax)))
        return float(tot)
    if kind.startswith("md_lower_bound"):
        s, x, p = where["s"], where["x"], mp.mpf(where["p"])
        return float(sum(mp.binomial(s, i) * p ** i * (1 - p) ** (s - i) for i in range(x, s + 1)))
    return None


# (k, pi, sketchSize) of every BASELINE.json configuration (stock and mathematical values), and App. C's sketch-size rows
TABLE_CASES = [(19, 0.85, 130), (19, 0.95, 40), (19, 0.95, 20), (19, 0.85, 310), (19, 0.85, 220), (19, 0.80, 498)]
SKETCH_SIZE_ROWS = [(19, 0.85, 5000, 100_000_000, 130), (19, 0.95, 10000, 18446744072414584320, 40), (19, 0.95, 10000, 3_000_000_000, 20),
                    (19, 0.85, 5000, 18446744072414584320, 310), (19, 0.85, 5000, 3_000_000_000, 220)]


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mashmap_amd import capi
    lib = capi.load()
    print("GSL boundary: integers of mm_stats.hpp (the product) against tests/gslcheck.py (scipy %s CDFs), and the margins of every comparison" % __import__("scipy").__version__)
    allM = Margins()
    for k, pi, s in TABLE_CASES:
        M = Margins()
        mine = [0] + [estimate_minimum_hits_relaxed(q, k, pi, CI, M) for q in range(1, s + 1)]
        theirs = [0] + [lib.mm_stat_min_hits_relaxed(q, k, pi) for q in range(1, s + 1)]
        cut = sketch_cutoffs(s, k, M)
        cut_p = capi.stat_sketch_cutoffs(s, k).tolist()
        print("k %d pi %.2f s %3d: minimum-hits table (Q.sketchSize 1..s) %s, sketchCutoffs %s" %
              (k, pi, s, "equal" if mine == theirs else "DIFFERENT", "equal" if cut == cut_p else "DIFFERENT"))
        for kind, (m, v, t, w) in sorted(M.best.items()):
            again = confirm_with_mpmath(kind, w)
            print("    %-42s closest: value %.12g vs %.12g, relative margin %.3g%s  at %s" %
                  (kind, v, t, m, "" if again is None else " (mpmath, 60 digits: %.12g)" % again, w))
            allM.see(kind, v, t, w)
    for k, pi, L, R, exp in SKETCH_SIZE_ROWS:
        M = Margins()
        got = recommended_sketch_size(k, pi, L, R, M)
        prod = lib.mm_stat_recommended_sketch_size(k, pi, L, R)
        m, v, t, w = M.best["recommendedSketchSize: pValue <= 1e-3"]
        print("recommendedSketchSize(k %d, pi %.2f, L %d, R %d) = %d (product %d, SURVEY App. C %d); closest p-value %.6g vs 1e-3, margin %.3g at s = %d" %
              (k, pi, L, R, got, prod, exp, v, m, w["s"]))
        allM.see("recommendedSketchSize", v, t, w)
    print("smallest relative margin of any comparison: %.3g (a CDF accurate to 1e-12 is %.0e times closer than that)" % (allM.floor(), allM.floor() / 1e-12))

// mashmap_amd/host/mm_stats.hpp -- host-side mirror of skch::Stat (src/map/include/map_stats.hpp:45-262)
// and of Map::setProbs (src/map/include/computeMap.hpp:178-258).
//
// Same function names, argument meaning and float/double mixing as the reference; the integers these
// produce (minimum hits, sketch cut-offs, recommended sketch size) are what the device kernels consume
// (include/mashmap_hip.h: mm_set_tables).  The reference delegates two tail probabilities to GNU GSL
// (gsl_cdf_binomial_Q, gsl_ran_hypergeometric_pdf / gsl_cdf_hypergeometric_P; GSL is not in this image).
// Here they are evaluated by direct log-space summation, which is exact to ~1e-13 for the n <= 1024
// these routines ever see.  They feed threshold comparisons only.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <atomic>
#include <functional>
#include <numeric>
#include <thread>
#include <vector>

namespace mmhost {
namespace Stat {

// lgamma(x + 1.0) for integer x: the tail sums below evaluate it millions of times at large sketch sizes (sketchSize^3 terms behind the
// minimum-hits table), always at small integers -- a table of the very same std::lgamma values, so every sum keeps its bits
inline double lgammaInt(unsigned x) {
  constexpr unsigned N = 1u << 15;
  static const std::vector<double> tab = [] { std::vector<double> t(N); for (unsigned i = 0; i < N; i++) t[i] = std::lgamma(i + 1.0); return t; }();
  return x < N ? tab[x] : std::lgamma(x + 1.0);
}
inline double lnChoose(unsigned n, unsigned m) { return lgammaInt(n) - lgammaInt(m) - lgammaInt(n - m); }

// P[X > k], X ~ Binomial(n, p)  (== gsl_cdf_binomial_Q(k, p, n))
inline double binomialUpperTail(unsigned k, double p, unsigned n) {
  if (k >= n) return 0.0;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return 1.0;
  const double lp = std::log(p), lq = std::log1p(-p);
  // sum the smaller side to avoid cancellation
  const double mean = (double)n * p;
  if ((double)k + 1.0 >= mean) {
    // At and beyond the mean the terms fall monotonically (t[i+1] / t[i] = (n - i) / (i + 1) * p / q <= np / (np + 1) < 1): once one is
    // below 2^-64 of the running sum, adding it -- or any later one -- leaves every bit of the sum as it is (half an ulp is 2^-54 of it;
    // the factor 2^10 between the two covers the ~1e-11 relative error of a computed term).  Same doubles as the full loop, which at
    // sketch sizes of several thousand spends its time on terms that underflow (round 4: the tables behind sketchSize 9 998 took a minute).
    double acc = 0.0;
    for (unsigned i = k + 1; i <= n; i++) {
      const double t = std::exp(lnChoose(n, i) + i * lp + (double)(n - i) * lq);
      acc += t;
      if (t < acc * 0x1p-64) break;
    }
    return acc > 1.0 ? 1.0 : acc;
  }
  double acc = 0.0;
  for (unsigned i = 0; i <= k; i++) acc += std::exp(lnChoose(n, i) + i * lp + (double)(n - i) * lq);
  return acc > 1.0 ? 0.0 : 1.0 - acc;
}

// == gsl_ran_hypergeometric_pdf(k, n1, n2, t)
inline double hypergeometricPdf(unsigned k, unsigned n1, unsigned n2, unsigned t) {
  if (t > n1 + n2) t = n1 + n2;
  if (k > n1 || k > t) return 0.0;
  if (t > n2 && k + n2 < t) return 0.0;
  return std::exp(lnChoose(n1, k) + lnChoose(n2, t - k) - lnChoose(n1 + n2, t));
}
// == gsl_cdf_hypergeometric_P(k, n1, n2, t)
inline double hypergeometricCdf(unsigned k, unsigned n1, unsigned n2, unsigned t) {
  if (k >= n1 || k >= t) return 1.0;
  double acc = 0.0;
  for (unsigned i = 0; i <= k; i++) acc += hypergeometricPdf(i, n1, n2, t);
  return acc > 1.0 ? 1.0 : acc;
}

inline float j2md(float j, int k) {                       // map_stats.hpp:45
  if (j == 0) return 1.0f;
  if (j == 1) return 0.0f;
  float mash_dist = 1 - std::pow(2 * j / (1 + j), 1.0 / k);
  return mash_dist;
}
inline float md2j(float d, int k) {                       // map_stats.hpp:63
  float sim = 1 - d;
  float jaccard = std::pow((double)sim, (double)k) / (2 - std::pow((double)sim, (double)k));
  return jaccard;
}
inline float md_lower_bound(float d, int s, int k, float ci) {   // map_stats.hpp:81 (GSL branch)
  float q2 = (1.0 - ci) / 2;
  int x = std::max(int(std::ceil(s * md2j(d, k))), 1);
  while (x <= s) {
    double cdf_complement = binomialUpperTail(x - 1, md2j(d, k), s);
    if (cdf_complement < q2) { x--; break; }
    x++;
  }
  float jaccard = float(x) / s;
  return j2md(jaccard, k);
}
inline int estimateMinimumHits(int s, int k, float perc_identity) {   // map_stats.hpp:122
  float mash_dist = 1.0 - perc_identity;
  float jaccard = md2j(mash_dist, k);
  return (int)std::ceil(1.0 * s * jaccard);
}
inline int estimateMinimumHitsRelaxed(int s, int k, float perc_identity, float confidence_interval) {   // map_stats.hpp:144
  const int first = estimateMinimumHits(s, k, perc_identity);
  int relaxed = first;
  for (int i = first; i >= 0; i--) {
    float jaccard = 1.0 * i / s;
    float d = j2md(jaccard, k);
    float d_lower = md_lower_bound(d, s, k, confidence_interval);
    float id_upper = 1.0 - d_lower;
    if (id_upper >= perc_identity) relaxed = i; else break;
  }
  return relaxed;
}
inline double estimate_pvalue(int s, int k, int alphabetSize, float identity, int64_t lengthQuery, uint64_t lengthReference,
                              float confidence_interval) {   // map_stats.hpp:181
  double kmerSpace = std::pow((double)alphabetSize, (double)k);
  double pX, pY; pX = pY = 1. / (1. + kmerSpace / lengthQuery);
  double r = pX * pY / (pX + pY - pX * pY);
  int x = estimateMinimumHitsRelaxed(s, k, identity, confidence_interval);
  double cdf_complement = (x == 0) ? 1.0 : binomialUpperTail(x - 1, r, s);
  return lengthReference * cdf_complement;
}
inline int64_t recommendedSketchSize(double pValue_cutoff, float confidence_interval, int k, int alphabetSize, float identity,
                                     int64_t segmentLength, uint64_t lengthReference) {   // map_stats.hpp:234
  int64_t lengthQuery = segmentLength - k;
  int optimalSketchSize;
  for (optimalSketchSize = 10; optimalSketchSize < lengthQuery; optimalSketchSize += 10)
    if (estimate_pvalue(optimalSketchSize, k, alphabetSize, identity, lengthQuery, lengthReference, confidence_interval) <= pValue_cutoff) break;
  return optimalSketchSize;
}

}  // namespace Stat

namespace fixed {                                          // map_parameters.hpp:86-102
constexpr double ss_table_max = 1000.0;
constexpr double pval_cutoff = 1e-3;
constexpr float confidence_interval = 0.95f;
constexpr float ANIDiff = 0.0f;
constexpr float ANIDiffConf = 0.999f;
}

// Map::sketchCutoffs as filled by Map::setProbs (computeMap.hpp:128,178-258).  The reference evaluates gsl_cdf_hypergeometric_P afresh at
// every step of every binary search (minutes at sketchSize >= 1000); here the running sums of each (ss, ss - ci, ci) distribution are
// kept per ci -- the same additions in the same order, so the same doubles -- and the cmax rows are independent (host threads).
inline std::vector<int> sketchCutoffs(int sketchSize, int kmerSize, float ANIDiff, float ANIDiffConf, bool stage1_topANI_filter, unsigned threads = 0) {
  std::vector<int> cut((size_t)(std::min<double>(sketchSize, fixed::ss_table_max) + 1), 1);
  if (!stage1_topANI_filter) return cut;
  const float deltaANI = ANIDiff;
  const float min_p = 1 - ANIDiffConf;
  const int ss = (int)std::min<double>(sketchSize, fixed::ss_table_max);
  if (threads == 0) threads = std::max(1u, std::min(std::thread::hardware_concurrency(), (unsigned)ss / 16u + 1u));   // starting a thread costs more than a row of a small table
  // probs[ci][y] = pdf(y; ss, ss - ci, ci); cum[ci][y] = probs[ci][0] + ... + probs[ci][y] (summed upwards, as hypergeometricCdf does)
  std::vector<std::vector<double>> probs(ss + 1, std::vector<double>(ss + 1)), cum(ss + 1);
  auto parallel = [&](int n, const std::function<void(int)>& fn) {
    std::atomic<int> next(0);
    auto work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads && (int)t < n; t++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
  };
  parallel(ss + 1, [&](int ci) {
    cum[ci].resize(ci + 1);
    double acc = 0.0;
    for (int y = 0; y <= ci; y++) { probs[ci][y] = Stat::hypergeometricPdf(y, ss, ss - ci, ci); acc += probs[ci][y]; cum[ci][y] = acc; }
  });
  // == Stat::hypergeometricCdf(k, ss, ss - ci, ci)
  auto cdf = [&](unsigned k, int ci) -> double {
    if (k >= (unsigned)ss || k >= (unsigned)ci) return 1.0;
    const double acc = cum[ci][k];
    return acc > 1.0 ? 1.0 : acc;
  };
  auto distDiff = [&](int cmax, int ci) {
    double prAbove = 0;
    for (int ymax = 0; ymax <= cmax; ymax++) {
      const double pymax = probs[cmax][ymax];
      const double yi_cutoff = deltaANI == 0 ? (double)ymax
          : std::floor(Stat::md2j(Stat::j2md((float)((double)ymax / ss), kmerSize) + deltaANI, kmerSize) * ss);
      double pi_acc = (yi_cutoff - 1) >= 0 ? cdf((unsigned)(yi_cutoff - 1), ci) : 0;
      pi_acc = 1 - pi_acc;
      prAbove += pymax * pi_acc;
      if (prAbove > min_p) return true;
    }
    return prAbove > min_p;
  };
  parallel(ss, [&](int i) {
    const int cmax = i + 1;
    // lowest ci in [0, ss) for which distDiff holds (the reference binary-searches a monotone predicate with std::upper_bound)
    int lo = 0, hi = ss;
    while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (distDiff(cmax, mid)) hi = mid; else lo = mid + 1; }
    cut[cmax] = lo == 0 ? 1 : lo;
  });
  return cut;
}

// estimateMinimumHitsRelaxed for every Q.sketchSize 0..sketchSize (computeMap.hpp:1144), on host threads (each entry is an O(q^2) search)
inline std::vector<int32_t> minHitsTable(int sketchSize, int kmerSize, float percentageIdentity, unsigned threads = 0) {
  std::vector<int32_t> mh((size_t)sketchSize + 1, 0);
  if (threads == 0) threads = std::max(1u, std::min(std::thread::hardware_concurrency(), (unsigned)sketchSize / 16u + 1u));
  std::atomic<int> next(sketchSize);                       // largest first: they take longest
  auto work = [&]() { for (int q = next.fetch_sub(1); q >= 1; q = next.fetch_sub(1)) mh[q] = Stat::estimateMinimumHitsRelaxed(q, kmerSize, percentageIdentity, fixed::confidence_interval); };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads && (int)t < sketchSize; t++) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  return mh;
}

// Integer tables that let doL2Mapping's best-first walk (computeMap.hpp:1182-1267) run without floating point:
//   accept[Qs * stride + shared]  = 1 when an L2 locus with `shared` of Qs sketch elements is reported (:1221-1222)
//   minIsz[Qs * stride + best]    = smallest L1 intersectionSize that survives the ANI cut-off (:1192-1202) once the best reported
//                                   locus so far shares `best` elements (Qs + 1: none does)
// Both are evaluated with the reference's own expressions (same float/double mixing), for every pair that can occur; rows are
// independent, so they are filled by `threads` host threads.  stride = sketchSize + 1.
inline void replayTables(int sketchSize, int k, float percentageIdentity, float ANIDiff, bool keep_low_pct_id, unsigned threads,
                         std::vector<uint8_t>& accept, std::vector<int16_t>& minIsz) {
  const size_t stride = (size_t)sketchSize + 1;
  accept.assign(stride * stride, 0); minIsz.assign(stride * stride, 0);
  if (threads < 1) threads = 1;
  threads = std::min(threads, (unsigned)sketchSize / 16u + 1u);
  std::atomic<int> next(1);
  auto work = [&]() {
    for (int Qs = next.fetch_add(1); Qs <= sketchSize; Qs = next.fetch_add(1)) {
      for (int shared = 0; shared <= Qs; shared++) {
        const float mash_dist = Stat::j2md(1.0 * shared / Qs, k);
        const float nucIdentity = (1 - mash_dist);
        bool ok = nucIdentity >= percentageIdentity;
        if (!ok && keep_low_pct_id) {
          const float ub = 1 - Stat::md_lower_bound(mash_dist, Qs, k, fixed::confidence_interval);
          ok = ub >= percentageIdentity;
        }
        accept[(size_t)Qs * stride + shared] = ok ? 1 : 0;
      }
      for (int best = 0; best <= Qs; best++) {
        const double bestJaccardNumerator = best;
        const double cutoff_ani = std::max(0.0, double((1 - Stat::j2md(bestJaccardNumerator / Qs, k)) - ANIDiff));
        const double cutoff_j = Stat::md2j(1 - cutoff_ani, k);
        // the smallest isz in [0, Qs + 1] for which double(isz) / Qs < cutoff_j no longer holds -- the reference counts up to it
        // (:1196-1200); the predicate is monotone in isz (a division by a positive constant is), so a bisection finds the same one
        int lo = 0, hi = Qs + 1;
        while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (double(mid) / Qs < cutoff_j) lo = mid + 1; else hi = mid; }
        minIsz[(size_t)Qs * stride + best] = (int16_t)lo;
      }
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads; t++) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
}

}  // namespace mmhost

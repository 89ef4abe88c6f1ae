/*
 * Minimal stand-in for the three GNU GSL entry points the reference links against
 * (GSL is an un-vendored, un-pinned system dependency of marbl/MashMap and is absent
 * from this image).  TEST INFRASTRUCTURE ONLY: used when compiling the reference
 * sources into oracle/_ref/ -- never linked into the product.
 *
 * Call sites in the reference:
 *   gsl_cdf_binomial_Q          src/map/include/map_stats.hpp:98, :213
 *   gsl_ran_hypergeometric_pdf  src/map/include/computeMap.hpp:194
 *   gsl_cdf_hypergeometric_P    src/map/include/computeMap.hpp:213
 *
 * Published GSL semantics restated here:
 *   binomial_Q(k; p, n)      = P[X > k], X ~ Bin(n, p) = I_p(k+1, n-k); 0 when k >= n
 *   hypergeometric_pdf(k; n1, n2, t) = C(n1,k) C(n2,t-k) / C(n1+n2,t), 0 off support
 *   hypergeometric_P(k; n1, n2, t)   = sum_{i<=k} pdf(i); 1 when k >= n1 or k >= t
 *
 * These feed only threshold comparisons in the reference (sketch size, minimum hits,
 * HG cut-offs).  "parity unpinned" at this boundary: see DESIGN.md.
 */
#pragma once
#include <cmath>

namespace gslshim {

/* continued fraction for the regularised incomplete beta function (modified Lentz) */
inline double beta_cf(double a, double b, double x) {
  const double tiny = 1e-300, eps = 1e-16;
  double c = 1.0;
  double d = 1.0 - (a + b) * x / (a + 1.0);
  if (std::fabs(d) < tiny) d = tiny;
  d = 1.0 / d;
  double f = d;
  for (int m = 1; m <= 20000; ++m) {
    const double m2 = 2.0 * m;
    /* even step */
    double num = m * (b - m) * x / ((a + m2 - 1.0) * (a + m2));
    d = 1.0 + num * d; if (std::fabs(d) < tiny) d = tiny; d = 1.0 / d;
    c = 1.0 + num / c; if (std::fabs(c) < tiny) c = tiny;
    f *= d * c;
    /* odd step */
    num = -(a + m) * (a + b + m) * x / ((a + m2) * (a + m2 + 1.0));
    d = 1.0 + num * d; if (std::fabs(d) < tiny) d = tiny; d = 1.0 / d;
    c = 1.0 + num / c; if (std::fabs(c) < tiny) c = tiny;
    const double delta = d * c;
    f *= delta;
    if (std::fabs(delta - 1.0) < eps) break;
  }
  return f;
}

/* I_x(a, b) */
inline double beta_inc(double a, double b, double x) {
  if (x <= 0.0) return 0.0;
  if (x >= 1.0) return 1.0;
  const double lnfront = std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b)
                       + a * std::log(x) + b * std::log1p(-x);
  const double front = std::exp(lnfront);
  if (x < (a + 1.0) / (a + b + 2.0)) return front * beta_cf(a, b, x) / a;
  return 1.0 - front * beta_cf(b, a, 1.0 - x) / b;
}

inline double ln_choose(unsigned n, unsigned m) {
  return std::lgamma(n + 1.0) - std::lgamma(m + 1.0) - std::lgamma((double)n - m + 1.0);
}

}  // namespace gslshim

static inline double gsl_cdf_binomial_Q(unsigned k, double p, unsigned n) {
  if (k >= n) return 0.0;
  return gslshim::beta_inc(k + 1.0, (double)n - k, p);
}

static inline double gsl_ran_hypergeometric_pdf(unsigned k, unsigned n1, unsigned n2, unsigned t) {
  if (t > n1 + n2) t = n1 + n2;
  if (k > n1 || k > t) return 0.0;
  if (t > n2 && k + n2 < t) return 0.0;
  return std::exp(gslshim::ln_choose(n1, k) + gslshim::ln_choose(n2, t - k)
                  - gslshim::ln_choose(n1 + n2, t));
}

static inline double gsl_cdf_hypergeometric_P(unsigned k, unsigned n1, unsigned n2, unsigned t) {
  if (k >= n1 || k >= t) return 1.0;
  double acc = 0.0;
  for (unsigned i = 0; i <= k; ++i) acc += gsl_ran_hypergeometric_pdf(i, n1, n2, t);
  return acc > 1.0 ? 1.0 : acc;
}

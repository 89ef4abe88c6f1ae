/*
 * oracle/oracle.cpp -- CPU restatement of marbl/MashMap v3.1.3's sketch + L1/L2 hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Written from the reference's *behaviour*; every
 * routine cites the reference file:line it restates (paths relative to /root/reference/).
 * Parity: pinned against the real reference (oracle/_ref) by tests/test_oracle_vs_ref.py and
 * against tests/golden/*.json.
 */
#include "oracle.h"
#include "gsl_shim/gsl/gsl_cdf.h"   /* GSL is absent from the image; same stand-in the _ref build uses */

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace {

/* ------------------------------------------------------------------------------------------
 * a3  MurmurHash3_x64_128, low 64 bits, seed 42     (murmur3.h:226-303, commonFunc.hpp:37,138)
 * ---------------------------------------------------------------------------------------- */
inline uint64_t rol64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t avalanche(uint64_t v) {            /* murmur3.h:57 fmix64 */
  v ^= v >> 33; v *= 0xff51afd7ed558ccdULL;
  v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL;
  v ^= v >> 33; return v;
}
const uint64_t MC1 = 0x87c37b91114253d5ULL, MC2 = 0x4cf5ad432745937fULL;
inline uint64_t mix_k1(uint64_t k) { k *= MC1; k = rol64(k, 31); k *= MC2; return k; }
inline uint64_t mix_k2(uint64_t k) { k *= MC2; k = rol64(k, 33); k *= MC1; return k; }

uint64_t murmur_lo64(const uint8_t* p, int len, uint32_t seed) {
  uint64_t h1 = seed, h2 = seed;
  const int nb = len / 16;
  for (int b = 0; b < nb; b++) {
    uint64_t k1, k2;
    std::memcpy(&k1, p + 16 * b, 8); std::memcpy(&k2, p + 16 * b + 8, 8);   /* little-endian block read */
    h1 ^= mix_k1(k1); h1 = rol64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    h2 ^= mix_k2(k2); h2 = rol64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t* t = p + 16 * nb;
  const int rem = len & 15;
  uint64_t k1 = 0, k2 = 0;
  for (int i = rem - 1; i >= 8; i--) k2 |= (uint64_t)t[i] << (8 * (i - 8));
  for (int i = std::min(rem, 8) - 1; i >= 0; i--) k1 |= (uint64_t)t[i] << (8 * i);
  if (rem > 8) h2 ^= mix_k2(k2);
  if (rem > 0) h1 ^= mix_k1(k1);
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = avalanche(h1); h2 = avalanche(h2);
  h1 += h2;                                         /* out[0]; out[1] (= h2 + h1) is discarded by getHash */
  return h1;
}
const uint32_t SEED = 42;
inline uint64_t kmer_hash(const char* s, int k) { return murmur_lo64((const uint8_t*)s, k, SEED); }

/* a2  reverse complement of one k-mer; non-ACGT bytes are copied unchanged (commonFunc.hpp:50) */
void revcomp(const char* src, char* dst, int n) {
  for (int i = 0; i < n; i++) {
    char c = src[i];
    c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
    dst[n - 1 - i] = c;
  }
}

/* a1  upper-case, then anything not in {A,C,G,T} becomes 'N' (commonFunc.hpp:75-107).
 *     The reference indexes a 127-entry table with a plain char (UB for bytes >= 127); on
 *     every byte < 127 the table says "invalid unless A/C/G/T".  We apply that rule to all bytes. */
void normalise(char* s, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    char c = s[i];
    if (c > 96 && c < 123) c -= 32;
    if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) c = 'N';
    s[i] = c;
  }
}

struct KmerRec { uint64_t h; int32_t pos; int8_t st; };

/* ------------------------------------------------------------------------------------------
 * a4  sketchSequence (commonFunc.hpp:183-288), restated order-free:
 *     valid k-mer = no 'N' inside it (initial scan :207-215 + running counter :220-223,272-275)
 *     and fwd hash != rc hash (:234); canonical = min, strand = fwd<rc ? +1 : -1 (:237-240);
 *     output the s smallest distinct canonical hashes ascending with first/last position and
 *     sign of the summed strand (:250,267-268,278-286).
 * ---------------------------------------------------------------------------------------- */
std::vector<orc_minmer> sketch_sequence(std::string seq, int k, int s, int seqId) {
  const int len = (int)seq.size();
  normalise(&seq[0], len);
  std::vector<KmerRec> recs;
  std::vector<char> rc(k);
  int lastN = -1;
  for (int j = 0; j < std::min(k - 1, len); j++) if (seq[j] == 'N') lastN = j;
  for (int i = 0; i + k <= len; i++) {
    if (seq[i + k - 1] == 'N') lastN = i + k - 1;
    if (lastN >= i) continue;                       /* an N inside [i, i+k) */
    const uint64_t f = kmer_hash(&seq[i], k);
    revcomp(&seq[i], rc.data(), k);
    const uint64_t b = kmer_hash(rc.data(), k);
    if (f == b) continue;
    recs.push_back(KmerRec{std::min(f, b), i, (int8_t)(f < b ? 1 : -1)});
  }
  std::sort(recs.begin(), recs.end(), [](const KmerRec& a, const KmerRec& b) {
    return a.h != b.h ? a.h < b.h : a.pos < b.pos; });
  std::vector<orc_minmer> out;
  for (size_t i = 0; i < recs.size() && (int)out.size() < s;) {
    size_t j = i; int sum = 0;
    while (j < recs.size() && recs[j].h == recs[i].h) { sum += recs[j].st; j++; }
    /* the reference accumulates in an int16 (strand_t); wrap the same way */
    const int16_t acc = (int16_t)sum;
    out.push_back(orc_minmer{recs[i].h, recs[i].pos, recs[j - 1].pos, seqId,
                             (int16_t)(acc > 0 ? 1 : (acc == 0 ? 0 : -1)), 0});
    i = j;
  }
  return out;
}

/* ------------------------------------------------------------------------------------------
 * a5  addMinmers (commonFunc.hpp:302-570).  This restatement keeps the reference's event
 *     sequence (departure :376-410, arrival :417-445, eviction/refill :455-505, final flush
 *     :509-520, clean-up :523-568) because the order in which records are *emitted* decides
 *     how std::sort (:558, not stable) arranges records with equal (wpos, wpos_end).
 * ---------------------------------------------------------------------------------------- */
struct Occ { int32_t pos; int16_t st; };
struct OpenMinmer {
  int32_t wpos, wpos_end; int16_t strandSum;
  std::deque<Occ> occ;                              /* occurrences of the hash currently counted */
};

std::vector<orc_minmer> add_minmers(std::string seq, int k, int w, int s, int seqId) {
  const int len = (int)seq.size();
  normalise(&seq[0], len);
  std::vector<orc_minmer> out;
  std::deque<KmerRec> inWindow;                     /* valid k-mers of the current window, oldest first (:315) */
  std::map<uint64_t, OpenMinmer> sketch;            /* the <= s smallest distinct hashes (:321-322) */
  /* pending k-mers not in the sketch: min-heap on (hash, pos) with lazy expiry (:319-323) */
  auto worse = [](const KmerRec& a, const KmerRec& b) { return std::tie(a.h, a.pos) > std::tie(b.h, b.pos); };
  std::vector<KmerRec> pending;
  std::vector<char> rc(k);
  int nCountdown = 0;                               /* no initial-N scan here (:334), unlike sketchSequence */

  auto emit = [&](uint64_t h, const OpenMinmer& m, int32_t end) {
    out.push_back(orc_minmer{h, m.wpos, end, seqId, m.strandSum, 0});
  };

  for (int i = 0; i + k <= len; i++) {
    const int W = i + k - w;                        /* window whose last k-mer is i (:341) */
    if ((int64_t)pending.size() > 2 * (int64_t)w) { /* bulk purge (:344-354) */
      pending.erase(std::remove_if(pending.begin(), pending.end(), [W](const KmerRec& r) { return r.pos < W; }),
                    pending.end());
      std::make_heap(pending.begin(), pending.end(), worse);
    }
    const uint64_t f = kmer_hash(&seq[i], k);
    revcomp(&seq[i], rc.data(), k);
    const uint64_t b = kmer_hash(rc.data(), k);
    const uint64_t h = std::min(f, b);
    const int16_t st = f < b ? 1 : -1;

    /* (i) the k-mer that slid out of the window (:376-410) */
    if (!inWindow.empty() && inWindow.front().pos < W) {
      const KmerRec gone = inWindow.front();
      if (!sketch.empty() && gone.h <= std::prev(sketch.end())->first) {
        auto it = sketch.find(gone.h);
        OpenMinmer& m = it->second;
        if (m.occ.size() == 1) {
          emit(gone.h, m, W);
          sketch.erase(it);
        } else {
          if (m.strandSum - gone.st == 0 || m.strandSum == 0) {
            emit(gone.h, m, W);
            m.wpos = W; m.wpos_end = -1;
          }
          m.strandSum -= gone.st;
          m.occ.pop_front();
        }
      }
      inWindow.pop_front();
    }

    /* (ii) the arriving k-mer (:412-449) */
    if (seq[i + k - 1] == 'N') nCountdown = k;
    if (f != b && nCountdown == 0) {
      inWindow.push_back(KmerRec{h, i, (int8_t)st});
      auto it = sketch.find(h);
      if (it != sketch.end()) {
        OpenMinmer& m = it->second;
        m.occ.push_back(Occ{i, st});
        if (m.strandSum + st == 0 || m.strandSum == 0) {
          emit(h, m, W);
          m.wpos = W; m.wpos_end = -1;
        }
        m.strandSum += st;
      } else {
        pending.push_back(KmerRec{h, i, (int8_t)st});
        std::push_heap(pending.begin(), pending.end(), worse);
      }
    }
    if (nCountdown > 0) nCountdown--;

    /* (iii) keep the sketch at the s smallest distinct hashes of window W (:455-505) */
    if (W >= 0) {
      while (!pending.empty() && pending.front().pos < W) { std::pop_heap(pending.begin(), pending.end(), worse); pending.pop_back(); }
      if (!sketch.empty() && !pending.empty() && (int)sketch.size() == s &&
          pending.front().h < std::prev(sketch.end())->first) {
        auto last = std::prev(sketch.end());
        emit(last->first, last->second, W);
        for (const Occ& o : last->second.occ)
          if (o.pos > W) { pending.push_back(KmerRec{last->first, o.pos, (int8_t)o.st}); std::push_heap(pending.begin(), pending.end(), worse); }
        sketch.erase(last);
      }
      while (!pending.empty() && (int)sketch.size() < s) {
        if (pending.front().pos < W) {              /* single lazy pop, as in the reference (:489-493) */
          std::pop_heap(pending.begin(), pending.end(), worse); pending.pop_back();
          if (pending.empty()) break;               /* the reference would read front() of an empty heap here */
        }
        const uint64_t nh = pending.front().h;
        OpenMinmer& m = sketch[nh];
        m.wpos = W; m.wpos_end = -1; m.strandSum = 0; m.occ.clear();
        while (!pending.empty() && pending.front().h == nh) {
          m.occ.push_back(Occ{pending.front().pos, pending.front().st});
          m.strandSum += pending.front().st;
          std::pop_heap(pending.begin(), pending.end(), worse); pending.pop_back();
        }
      }
    }
  }

  /* final flush in ascending hash order, closed at len-k+1 (:509-520) */
  {
    uint64_t rank = 1;
    for (auto it = sketch.begin(); it != sketch.end() && rank <= (uint64_t)s; ++it, ++rank)
      if (it->second.wpos != -1) emit(it->first, it->second, len - k + 1);
  }
  /* drop malformed / empty runs (:523-528) */
  out.erase(std::remove_if(out.begin(), out.end(), [](const orc_minmer& m) {
    return m.wpos < 0 || m.wpos_end < 0 || m.wpos == m.wpos_end; }), out.end());
  /* strand: sum<0 -> REV(-1), otherwise FWD(+1) (zero maps to FWD) (:534); split runs longer than w (:535-555) */
  std::vector<orc_minmer> pieces;
  for (auto& m : out) {
    m.strand = m.strand < 0 ? -1 : 1;
    if (m.wpos_end > m.wpos + w) {
      const int nchunk = (int)std::ceil(float(m.wpos_end - m.wpos) / float(w));   /* float arithmetic as in :536 */
      for (int c = 0; c < nchunk; c++)
        pieces.push_back(orc_minmer{m.hash, m.wpos + c * w, std::min(m.wpos + c * w + w, m.wpos_end), m.seqId, m.strand, 0});
    }
  }
  out.erase(std::remove_if(out.begin(), out.end(), [w](const orc_minmer& m) { return m.wpos_end - m.wpos > w; }), out.end());
  out.insert(out.end(), pieces.begin(), pieces.end());
  /* same algorithm + same comparator as :558 so that ties fall the same way */
  std::sort(out.begin(), out.end(), [](const orc_minmer& l, const orc_minmer& r) {
    return std::tie(l.wpos, l.wpos_end) < std::tie(r.wpos, r.wpos_end); });
  out.erase(std::unique(out.begin(), out.end(), [](const orc_minmer& l, const orc_minmer& r) {
    return l.wpos == r.wpos && l.hash == r.hash; }), out.end());
  return out;
}

/* ------------------------------------------------------------------------------------------
 * a14  Stat:: (map_stats.hpp).  Float/double mixing is kept expression by expression.
 * ---------------------------------------------------------------------------------------- */
float j2md(float j, int k) {                         /* :45-55 */
  if (j == 0) return 1.0f;
  if (j == 1) return 0.0f;
  const float ratio = 2 * j / (1 + j);               /* float arithmetic */
  const float d = (float)(1 - std::pow((double)ratio, 1.0 / k));
  return d;
}
float md2j(float d, int k) {                         /* :63-68 */
  const float sim = 1 - d;
  const double p = std::pow((double)sim, (double)k);
  return (float)(p / (2 - p));
}
float md_lower_bound(float d, int s, int k, float ci) {   /* :81-112, GSL branch */
  const float q2 = (float)((1.0 - ci) / 2);
  int x = std::max((int)std::ceil((float)s * md2j(d, k)), 1);
  while (x <= s) {
    const double tail = gsl_cdf_binomial_Q(x - 1, md2j(d, k), s);
    if (tail < q2) { x--; break; }
    x++;
  }
  const float jac = float(x) / s;
  return j2md(jac, k);
}
int min_hits(int s, int k, float pi) {               /* :122-134 */
  const float md = (float)(1.0 - pi);
  const float jac = md2j(md, k);
  return (int)std::ceil(1.0 * s * jac);
}
int min_hits_relaxed(int s, int k, float pi, float ci) {  /* :144-169 */
  const int upper = min_hits(s, k, pi);
  int best = upper;
  for (int i = upper; i >= 0; i--) {
    const float jac = (float)(1.0 * i / s);
    const float d = j2md(jac, k);
    const float dlow = md_lower_bound(d, s, k, ci);
    const float idUpper = (float)(1.0 - dlow);
    if (idUpper >= pi) best = i; else break;
  }
  return best;
}
double estimate_pvalue(int s, int k, int alphabet, float pi, int64_t lenQuery, uint64_t lenRef, float ci) {  /* :181-219 */
  const double space = std::pow((double)alphabet, (double)k);
  const double pX = 1. / (1. + space / lenQuery), pY = pX;
  const double r = pX * pY / (pX + pY - pX * pY);
  const int x = min_hits_relaxed(s, k, pi, ci);
  const double tail = x == 0 ? 1.0 : gsl_cdf_binomial_Q(x - 1, r, s);
  return lenRef * tail;
}
int64_t recommended_sketch_size(double pcut, float ci, int k, int alphabet, float pi, int64_t segLength, uint64_t lenRef) {  /* :234-258 */
  const int64_t lenQuery = segLength - k;
  int ss;
  for (ss = 10; ss < lenQuery; ss += 10)
    if (estimate_pvalue(ss, k, alphabet, pi, lenQuery, lenRef, ci) <= pcut) break;
  return ss;
}
const float CONF_INTERVAL = 0.95f;                   /* map_parameters.hpp:97 */
const double PVAL_CUTOFF = 1e-3, SS_TABLE_MAX = 1000.0;   /* :95-96 */
const float ANI_DIFF = 0.0f, ANI_DIFF_CONF = 0.999f; /* :99-100 */

/* ------------------------------------------------------------------------------------------
 * Session = Sketch (winSketch.hpp) + Map parameters (computeMap.hpp)
 * ---------------------------------------------------------------------------------------- */
struct Contig { std::string name; int32_t len; };

struct Session {
  int k, segLength, sketchSize; float pi; int filterMode, flags; char delim; float kmerPct; int numMappings;
  std::vector<Contig> meta;
  std::vector<orc_minmer> index;                                   /* Sketch::minmerIndex (:102) */
  std::map<uint64_t, std::vector<orc_point>> lookup;               /* Sketch::minmerPosLookupIndex (:101) */
  std::set<uint64_t> frequent;                                     /* Sketch::frequentSeeds (:68) */
  int freqThreshold = std::numeric_limits<int>::max();             /* :65 */
  std::vector<int> cutoffs;                                        /* Map::sketchCutoffs (computeMap.hpp:109) */
  std::vector<int> refGroup;                                       /* Map::refIdGroup (:113) */

  bool hg() const { return flags & ORC_HG; }
  bool split() const { return !(flags & ORC_NOSPLIT); }
};

std::string prefix_of(const std::string& s, char c) { return s.substr(0, s.find_last_of(c)); }   /* computeMap.hpp:1170 */

/* a6  Sketch::index (winSketch.hpp:379-404) */
void build_lookup(Session& S) {
  for (const auto& mi : S.index) {
    auto& v = S.lookup[mi.hash];
    if (v.empty() || v.back().hash != mi.hash || v.back().pos != mi.wpos) {
      orc_point a; std::memset(&a, 0, sizeof a); a.pos = mi.wpos; a.hash = mi.hash; a.seqId = mi.seqId; a.side = 1;
      orc_point b = a; b.pos = mi.wpos_end; b.side = -1;
      v.push_back(a); v.push_back(b);
    } else {
      v.back().pos = mi.wpos_end;                    /* adjacent run of the same hash: extend the CLOSE point */
    }
  }
}

/* a7  computeFreqHist / computeFreqSeedSet / dropFreqSeedSet (winSketch.hpp:410-504) */
void frequency_filter(Session& S) {
  if (!S.lookup.empty()) {
    std::map<int, int> hist;
    for (auto& e : S.lookup) hist[(int)e.second.size()] += 1;
    const int64_t total = (int64_t)S.lookup.size();
    const int64_t toIgnore = (int64_t)(total * S.kmerPct / 100);   /* int64*float/int in float (:425) */
    int64_t sum = 0;
    for (auto it = hist.rbegin(); it != hist.rend(); ++it) {
      sum += it->second;
      if (sum < toIgnore) S.freqThreshold = it->first;
      else if (sum == toIgnore) { S.freqThreshold = it->first; break; }
      else break;
    }
  }
  for (auto& e : S.lookup) if ((int64_t)e.second.size() >= (int64_t)S.freqThreshold) S.frequent.insert(e.first);
  S.index.erase(std::remove_if(S.index.begin(), S.index.end(), [&](const orc_minmer& m) { return S.frequent.count(m.hash) != 0; }),
                S.index.end());
}

/* Map::setProbs (computeMap.hpp:178-258) */
void set_probs(Session& S) {
  const float deltaANI = ANI_DIFF;
  const float min_p = 1 - ANI_DIFF_CONF;
  const int ss = (int)std::min<double>(S.sketchSize, SS_TABLE_MAX);
  std::vector<std::vector<double>> pmf(ss + 1, std::vector<double>(ss + 1));
  for (int ci = 0; ci <= ss; ci++)
    for (double y = 0; y <= ci; y++) pmf[ci][(size_t)y] = gsl_ran_hypergeometric_pdf((unsigned)y, ss, ss - ci, ci);
  auto distDiff = [&](int cmax, int ci) {
    double pr = 0;
    for (double ymax = 0; ymax <= cmax; ymax++) {
      const double pymax = pmf[cmax][(size_t)ymax];
      const double cut = deltaANI == 0 ? ymax
          : std::floor(md2j(j2md((float)(ymax / ss), S.k) + deltaANI, S.k) * ss);
      double acc = (cut - 1) >= 0 ? gsl_cdf_hypergeometric_P((unsigned)(cut - 1), ss, ss - ci, ci) : 0;
      acc = 1 - acc;
      pr += pymax * acc;
      if (pr > min_p) return true;
    }
    return pr > min_p;
  };
  std::vector<int> range(ss + 1);
  std::iota(range.begin(), range.end(), 0);
  for (int cmax = 1; cmax <= ss; cmax++) {
    int ci = (int)std::distance(range.begin(),
        std::upper_bound(range.begin(), range.begin() + ss, false,
                         [&](bool, int c) { return distDiff(cmax, c); }));
    S.cutoffs[cmax] = ci;
    if (S.cutoffs[cmax] == 0) S.cutoffs[cmax] = 1;
  }
}

/* Map::setRefGroups / getRefGroup (computeMap.hpp:144-177) */
void set_ref_groups(Session& S) {
  int group = 0; size_t start = 0;
  while (start < S.meta.size()) {
    const std::string cur = prefix_of(S.meta[start].name, S.delim);
    size_t idx = start;
    while (idx < S.meta.size() && cur == prefix_of(S.meta[idx].name, S.delim)) S.refGroup[idx++] = group;
    group++; start = idx;
  }
}
int get_ref_group(const Session& S, const std::string& name) {
  const std::string q = prefix_of(name, S.delim);
  for (size_t i = 0; i < S.meta.size(); i++) if (q == prefix_of(S.meta[i].name, S.delim)) return S.refGroup[i];
  return -1;
}

/* ------------------------------------------------------------------------------------------
 * Per-fragment state (QueryMetaData, base_types.hpp:265)
 * ---------------------------------------------------------------------------------------- */
struct Frag {
  std::string seq; int32_t len, fullLen; int seqCounter; std::string name; int refGroup;
  std::vector<orc_minmer> sketch;   /* minmerTableQuery after frequent-seed removal */
  int sketchSize = 0; int rawSketchSize = 0; float kmerComplexity = 0;
};

/* a8  getSeedHits (computeMap.hpp:818-843) */
void seed_hits(const Session& S, Frag& Q) {
  Q.sketch = sketch_sequence(Q.seq, S.k, S.sketchSize, Q.seqCounter);
  Q.rawSketchSize = (int)Q.sketch.size();
  if (Q.sketch.empty()) { Q.sketchSize = 0; return; }
  const double maxHash01 = (double)((long double)Q.sketch.back().hash / std::numeric_limits<uint64_t>::max());
  Q.kmerComplexity = (float)((double(Q.sketch.size()) / maxHash01) / ((Q.len - S.k + 1) * 2));
  Q.sketch.erase(std::remove_if(Q.sketch.begin(), Q.sketch.end(), [&](const orc_minmer& m) { return S.frequent.count(m.hash) != 0; }),
                 Q.sketch.end());
  Q.sketchSize = (int)Q.sketch.size();
}

inline bool point_less(const orc_point& a, const orc_point& b) {   /* IntervalPoint::operator< (base_types.hpp:75) */
  return std::tie(a.seqId, a.pos, a.side) < std::tie(b.seqId, b.pos, b.side);
}

/* a9  getSeedIntervalPoints (computeMap.hpp:857-912): k-way merge of the per-hash point lists by
 *     (seqId, pos, side), with the skip_self / skip_prefix / lower_triangular filters (:891-896).
 *     Points comparing equal may come out in a different relative order than the reference's
 *     binary heap yields; nothing downstream looks at more than (seqId, pos, side) when
 *     windowLen == 0, and hash only as a counter key otherwise. */
void seed_interval_points(const Session& S, const Frag& Q, std::vector<orc_point>& pts) {
  if (Q.sketch.empty()) return;
  for (const auto& mi : Q.sketch) {
    auto it = S.lookup.find(mi.hash);
    if (it == S.lookup.end()) continue;
    for (const auto& p : it->second) {
      const Contig& ref = S.meta[p.seqId];
      if ((!(S.flags & ORC_SKIP_SELF) || Q.name != ref.name) &&
          (!(S.flags & ORC_SKIP_PREFIX) || S.refGroup[p.seqId] != Q.refGroup) &&
          (!(S.flags & ORC_LOWER_TRI) || Q.seqCounter > p.seqId))
        pts.push_back(p);
    }
  }
  std::stable_sort(pts.begin(), pts.end(), point_less);
}

/* a10  computeL1CandidateRegions (computeMap.hpp:916-1116) -- two sweeps, then the cluster join */
void l1_candidates(const Session& S, const Frag& Q, const orc_point* begin, const orc_point* end, int minimumHits,
                   std::vector<orc_l1>& l1out) {
  int overlap = 0, best = 0;
  std::vector<orc_l1> runs;
  const int windowLen = std::max<int32_t>(0, Q.len - S.segLength);
  const int clusterLen = S.segLength;
  std::unordered_map<uint64_t, int> openCount;
  auto retire_ok = [&](const orc_point* t, const orc_point* lead) {
    return (t->seqId == lead->seqId && t->pos <= lead->pos - windowLen) || t->seqId < lead->seqId; };

  if (S.hg()) {                                       /* pass 1 (:948-999) */
    const orc_point *trail = begin, *lead = begin;
    while (lead != end) {
      while (trail != end && retire_ok(trail, lead)) {
        if (trail->side == -1) {
          if (windowLen != 0) openCount[trail->hash]--;
          if (windowLen == 0 || openCount[trail->hash] == 0) overlap--;
        }
        trail++;
      }
      const int32_t cur = lead->pos;
      while (lead != end && lead->pos == cur) {
        if (lead->side == 1) {
          if (windowLen == 0 || openCount[lead->hash] == 0) overlap++;
          if (windowLen != 0) openCount[lead->hash]++;
        }
        lead++;
      }
      best = std::max(best, overlap);
    }
    if (best < minimumHits) return;
    minimumHits = std::max(S.cutoffs[int(std::min(best, Q.sketchSize) / std::max<double>(1, S.sketchSize / SS_TABLE_MAX))],
                           minimumHits);
  }
  openCount.clear();

  bool inRun = false;                                 /* pass 2 (:1009-1098) */
  orc_l1 cur{0, 0, 0, 0};
  const orc_point *trail = begin, *lead = begin;
  overlap = 0; int prevOverlap = 0;
  int32_t prevSeq = 0, prevPos = 0;                   /* uninitialised in the reference until first use */
  int32_t curSeq = lead->seqId, curPos = lead->pos;
  while (lead != end) {
    prevOverlap = overlap;
    while (trail != end && retire_ok(trail, lead)) {
      if (trail->side == -1) {
        if (windowLen != 0) openCount[trail->hash]--;
        if (windowLen == 0 || openCount[trail->hash] == 0) overlap--;
      }
      trail++;
    }
    if (lead->pos != curPos) { prevSeq = curSeq; prevPos = curPos; curSeq = lead->seqId; curPos = lead->pos; }
    while (lead != end && lead->pos == curPos) {
      if (lead->side == 1) {
        if (windowLen == 0 || openCount[lead->hash] == 0) overlap++;
        if (windowLen != 0) openCount[lead->hash]++;
      }
      lead++;
    }
    if (prevOverlap >= minimumHits) {
      if (cur.seqId != prevSeq && inRun) { runs.push_back(cur); cur = orc_l1{0, 0, 0, 0}; inRun = false; }
      if (!inRun) {
        cur.rangeStartPos = prevPos - windowLen; cur.rangeEndPos = prevPos - windowLen;
        cur.seqId = prevSeq; cur.intersectionSize = prevOverlap; inRun = true;
      } else {                                        /* stage2_full_scan is always true (parseCmdArgs.hpp:590) */
        cur.intersectionSize = std::max(cur.intersectionSize, prevOverlap);
        cur.rangeEndPos = prevPos - windowLen;
      }
    } else {
      if (inRun) { runs.push_back(cur); cur = orc_l1{0, 0, 0, 0}; }
      inRun = false;
    }
  }
  if (inRun) runs.push_back(cur);

  for (const auto& r : runs) {                        /* join (:1102-1115) */
    if (l1out.empty() || r.seqId != l1out.back().seqId || r.rangeStartPos > l1out.back().rangeEndPos + clusterLen) {
      l1out.push_back(r);
    } else {
      l1out.back().rangeEndPos = r.rangeEndPos;
      l1out.back().intersectionSize = std::max(r.intersectionSize, l1out.back().intersectionSize);
    }
  }
}

/* a11  doL1Mapping (computeMap.hpp:1130-1166); returns the minimumHits it used */
int l1_mapping(const Session& S, Frag& Q, std::vector<orc_point>& pts, std::vector<orc_l1>& l1) {
  seed_hits(S, Q);
  if (Q.sketchSize == 0 || Q.kmerComplexity < 0.0f /* kmerComplexityThreshold default (parseCmdArgs.hpp:565) */) return 0;
  seed_interval_points(S, Q, pts);
  const int minimumHits = min_hits_relaxed(Q.sketchSize, S.k, S.pi, CONF_INTERVAL);
  size_t b = 0;
  while (b < pts.size()) {
    size_t e = pts.size();
    if (S.flags & ORC_SKIP_PREFIX) {
      const int g = S.refGroup[pts[b].seqId];
      e = b; while (e < pts.size() && S.refGroup[pts[e].seqId] == g) e++;
    }
    l1_candidates(S, Q, pts.data() + b, pts.data() + e, minimumHits, l1);
    b = e;
  }
  return minimumHits;
}

/* ------------------------------------------------------------------------------------------
 * a13  SlideMapper (slidingMap.hpp:28-212), incremental form as in the reference
 * ---------------------------------------------------------------------------------------- */
struct Slide {
  struct Cell { uint64_t h; int16_t qStrand; int16_t vote; unsigned nBefore; bool active; };
  std::vector<Cell> cells;            /* [0] is a sentinel with hash 0 (:83, :106-121) */
  int pivot; size_t pivRank; int S;
  int shared = 0, votes = 0, inter = 0;
  explicit Slide(const Frag& Q) : cells(Q.sketchSize + 1, Cell{0, 0, 0, 0, false}), S(Q.sketchSize) {
    int idx = 1;
    for (const auto& m : Q.sketch) cells[idx++] = Cell{m.hash, m.strand, 0, 1, false};
    pivot = (int)cells.size() - 1; pivRank = cells.size() - 1;
  }
  int locate(uint64_t h) const {      /* lower_bound over cells[1..] (:128-131) */
    int lo = 1, hi = (int)cells.size();
    while (lo < hi) { int mid = (lo + hi) / 2; if (cells[mid].h < h) lo = mid + 1; else hi = mid; }
    return lo;
  }
  void insert(uint64_t h, int16_t rStrand) {          /* :125-165 */
    const int j = locate(h);
    if (j == (int)cells.size()) return;
    Cell& c = cells[j];
    if (c.h == h) {
      c.active = true; c.vote += c.qStrand * rStrand; inter++;
      if (c.h <= cells[pivot].h) { shared++; votes += c.vote; }
    } else {
      c.nBefore++;
      if (c.h <= cells[pivot].h) pivRank++;
      if (pivRank > (size_t)S) {
        shared -= cells[pivot].active; votes -= cells[pivot].vote; pivRank -= cells[pivot].nBefore; pivot--;
      }
    }
  }
  void remove(uint64_t h) {                            /* :171-211 */
    const int j = locate(h);
    if (j == (int)cells.size()) return;
    Cell& c = cells[j];
    if (c.h == h) {
      if (c.h <= cells[pivot].h) { shared--; votes -= c.vote; }
      c.active = false; c.vote = 0; inter--;
    } else {
      c.nBefore--;
      if (c.h <= cells[pivot].h) pivRank--;
      if (pivot + 1 != (int)cells.size() && pivRank + cells[pivot + 1].nBefore <= (size_t)S) {
        pivot++; shared += cells[pivot].active; votes += cells[pivot].vote; pivRank += cells[pivot].nBefore;
      }
    }
  }
};

inline bool minmer_less(const orc_minmer& a, const orc_minmer& b) {   /* MinmerInfo::operator< (base_types.hpp:59) */
  return std::tie(a.seqId, a.wpos) < std::tie(b.seqId, b.wpos);
}

/* a13  computeL2MappedRegions (computeMap.hpp:1276-1451) */
void l2_regions(const Session& S, const Frag& Q, const orc_l1& cand, std::vector<orc_l2>& out) {
  const auto& idx = S.index;
  orc_minmer probe{0, cand.rangeStartPos - S.segLength - 1, 0, cand.seqId, 0, 0};
  size_t it = std::lower_bound(idx.begin(), idx.end(), probe, minmer_less) - idx.begin();
  const size_t n = idx.size();
  std::vector<orc_minmer> open;                      /* min-heap on wpos_end (:1296-1300) */
  auto later = [](const orc_minmer& l, const orc_minmer& r) { return l.wpos_end > r.wpos_end; };
  const int windowLen = std::max<int32_t>(0, Q.len - S.segLength);
  std::unordered_map<uint64_t, int> openCount;
  Slide slide(Q);
  int bestShared = 1; bool inRun = false;
  orc_l2 cur{0, 0, 0, 0, 0, 0};
  /* position of the record after `i` if it lies in the same contig, else of `i` itself (:1387-1390);
     the reference dereferences one-past-the-end for the very last record -- we treat that as "no next". */
  auto next_wpos = [&](size_t i) { return (i + 1 < n && idx[i + 1].seqId == idx[i].seqId) ? idx[i + 1].wpos : idx[i].wpos; };
  auto close_run = [&]() {
    if (out.empty() || out.back().optimalEnd + S.segLength < cur.optimalStart) out.push_back(cur);
    else { out.back().optimalEnd = cur.optimalEnd; out.back().meanOptimalPos = (out.back().optimalStart + out.back().optimalEnd) / 2; }
  };

  while (it < n && idx[it].seqId == cand.seqId && idx[it].wpos < cand.rangeStartPos) {      /* pre-load (:1323-1338) */
    if (idx[it].wpos_end > cand.rangeStartPos) {
      if (windowLen > 0) openCount[idx[it].hash]++;
      if (windowLen == 0 || openCount[idx[it].hash] == 1) {
        open.push_back(idx[it]); std::push_heap(open.begin(), open.end(), later);
        slide.insert(idx[it].hash, idx[it].strand);
      }
    }
    it++;
  }
  while (it < n && idx[it].seqId == cand.seqId && idx[it].wpos <= cand.rangeEndPos + windowLen) {   /* slide (:1340-1434) */
    const int prevVotes = slide.votes;
    while (!open.empty() && open.front().wpos_end <= idx[it].wpos - windowLen) {
      if (windowLen > 0) openCount[open.front().hash]--;
      if (windowLen == 0 || openCount[open.front().hash] == 0) {
        slide.remove(open.front().hash);
        std::pop_heap(open.begin(), open.end(), later); open.pop_back();
      }
      /* (with windowLen > 0 and a still-positive count the reference spins here forever; unreachable in split mode) */
    }
    if (windowLen > 0) openCount[idx[it].hash]++;
    if (windowLen == 0 || openCount[idx[it].hash] == 1) {
      slide.insert(idx[it].hash, idx[it].strand);
      open.push_back(idx[it]); std::push_heap(open.begin(), open.end(), later);
    } else { it++; continue; }

    if (slide.shared > bestShared) {
      out.clear();
      inRun = true; bestShared = slide.shared; cur.sharedSketchSize = slide.shared;
      cur.optimalStart = idx[it].wpos;
      cur.optimalEnd = next_wpos(it) - windowLen;
    } else if (slide.shared == bestShared) {
      if (!inRun) { cur.sharedSketchSize = slide.shared; cur.optimalStart = idx[it].wpos - windowLen; }
      inRun = true;
      cur.optimalEnd = next_wpos(it) - windowLen;
    } else {
      if (inRun) {
        cur.optimalEnd = next_wpos(it) - windowLen;
        cur.meanOptimalPos = (cur.optimalStart + cur.optimalEnd) / 2;
        cur.seqId = idx[it].seqId;
        cur.strand = prevVotes >= 0 ? 1 : -1;
        close_run();
        cur = orc_l2{0, 0, 0, 0, 0, 0};
      }
      inRun = false;
    }
    it++;
  }
  if (inRun) {                                         /* :1435-1450 */
    cur.meanOptimalPos = (cur.optimalStart + cur.optimalEnd) / 2;
    cur.seqId = idx[it - 1].seqId;
    cur.strand = slide.votes >= 0 ? 1 : -1;
    close_run();
  }
}

/* a12  doL2Mapping (computeMap.hpp:1182-1267) on [l1b, l1e) */
void l2_mapping(const Session& S, const Frag& Q, std::vector<orc_l1>& l1, size_t l1b, size_t l1e, std::vector<orc_mapping>& outMaps) {
  auto byIntersection = [](const orc_l1& a, const orc_l1& b) { return a.intersectionSize < b.intersectionSize; };
  std::vector<orc_l2> loci;
  double bestNumerator = 0;
  size_t pos = l1b;
  while (pos != l1e) {
    orc_l1& cand = l1[pos];
    if (S.hg()) {
      const double cutoffAni = std::max(0.0, double((1 - j2md((float)(bestNumerator / Q.sketchSize), S.k)) - ANI_DIFF));
      const double cutoffJ = md2j((float)(1 - cutoffAni), S.k);
      if (double(cand.intersectionSize) / Q.sketchSize < cutoffJ) break;
    }
    loci.clear();
    l2_regions(S, Q, cand, loci);
    for (const auto& l2 : loci) {
      const float md = j2md((float)(1.0 * l2.sharedSketchSize / Q.sketchSize), S.k);
      const float ident = 1 - md;
      const float identUpper = 1 - md_lower_bound(md, Q.sketchSize, S.k, CONF_INTERVAL);
      const bool keepLow = !(S.flags & ORC_DROP_LOW_ID);
      if ((keepLow && identUpper >= S.pi) || ident >= S.pi) {
        bestNumerator = std::max<double>(bestNumerator, l2.sharedSketchSize);
        orc_mapping r; std::memset(&r, 0, sizeof r);
        r.queryLen = Q.len; r.refStartPos = l2.meanOptimalPos; r.refEndPos = l2.meanOptimalPos + Q.len;
        r.queryStartPos = 0; r.queryEndPos = Q.len; r.refSeqId = l2.seqId; r.querySeqId = Q.seqCounter;
        r.nucIdentity = ident; r.nucIdentityUpperBound = identUpper; r.sketchSize = Q.sketchSize;
        r.conservedSketches = l2.sharedSketchSize;
        r.blockLength = std::max(r.refEndPos - r.refStartPos, r.queryEndPos - r.queryStartPos);
        r.approxMatches = (int)std::round(r.nucIdentity * r.blockLength / 100.0);
        r.strand = l2.strand; r.kmerComplexity = Q.kmerComplexity;
        outMaps.push_back(r);
      }
    }
    if (S.hg()) { std::pop_heap(l1.begin() + l1b, l1.begin() + l1e, byIntersection); l1e--; }
    else pos++;
  }
}

/* mapSingleQueryFrag (computeMap.hpp:756-815) */
void map_fragment(const Session& S, Frag& Q, std::vector<orc_point>& pts, std::vector<orc_l1>& l1, std::vector<orc_mapping>& maps) {
  l1_mapping(S, Q, pts, l1);
  if (l1.empty()) return;
  auto byIntersection = [](const orc_l1& a, const orc_l1& b) { return a.intersectionSize < b.intersectionSize; };
  size_t b = 0;
  while (b < l1.size()) {
    size_t e = l1.size();
    if (S.flags & ORC_SKIP_PREFIX) {
      const int g = S.refGroup[l1[b].seqId];
      e = b; while (e < l1.size() && S.refGroup[l1[e].seqId] == g) e++;
    }
    if (S.hg()) std::make_heap(l1.begin() + b, l1.begin() + e, byIntersection);
    l2_mapping(S, Q, l1, b, e, maps);
    b = e;
  }
  std::sort(maps.begin(), maps.end(), [](const orc_mapping& a, const orc_mapping& b) {
    return std::tie(a.refSeqId, a.refStartPos) < std::tie(b.refSeqId, b.refStartPos); });
}

}  // namespace

/* ============================================================================================ */
extern "C" {

uint64_t orc_get_hash(const char* s, int len) { return kmer_hash(s, len); }
void orc_normalise(char* seq, int64_t len) { normalise(seq, len); }

int orc_sketch_sequence(const char* seq, int len, int k, int s, int seqId, orc_minmer* out, int cap) {
  auto v = sketch_sequence(std::string(seq, seq + len), k, s, seqId);
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
  return (int)v.size();
}
int64_t orc_add_minmers(const char* seq, int len, int k, int w, int s, int seqId, orc_minmer* out, int64_t cap) {
  auto v = add_minmers(std::string(seq, seq + len), k, w, s, seqId);
  for (size_t i = 0; i < v.size() && (int64_t)i < cap; i++) out[i] = v[i];
  return (int64_t)v.size();
}

float orc_j2md(float j, int k) { return j2md(j, k); }
float orc_md2j(float d, int k) { return md2j(d, k); }
float orc_md_lower_bound(float d, int s, int k, float ci) { return md_lower_bound(d, s, k, ci); }
int orc_min_hits(int s, int k, float pi) { return min_hits(s, k, pi); }
int orc_min_hits_relaxed(int s, int k, float pi) { return min_hits_relaxed(s, k, pi, CONF_INTERVAL); }
int64_t orc_recommended_sketch_size(int k, float pi, int64_t segLength, uint64_t refSize) {
  return recommended_sketch_size(PVAL_CUTOFF, CONF_INTERVAL, k, 4, pi, segLength, refSize);
}

void* orc_session_new(int k, int segLength, int sketchSize, float pi, int filterMode, int flags,
                      char prefixDelim, float kmerPctThreshold, int numMappings) {
  auto* S = new Session();
  S->k = k; S->segLength = segLength; S->sketchSize = sketchSize; S->pi = pi; S->filterMode = filterMode;
  S->flags = flags; S->delim = (flags & ORC_SKIP_PREFIX) ? prefixDelim : '\0'; S->kmerPct = kmerPctThreshold;
  S->numMappings = numMappings;
  return S;
}
void orc_session_add_contig(void* h, const char* name, const char* seq, int len) {
  auto* S = (Session*)h;
  const int seqId = (int)S->meta.size();
  S->meta.push_back(Contig{name, len});
  if (len < S->k) return;                            /* winSketch.hpp:194 */
  auto v = add_minmers(std::string(seq, seq + len), S->k, S->segLength, S->sketchSize, seqId);
  S->index.insert(S->index.end(), v.begin(), v.end());
}
/* replaces minmerIndex before orc_session_finalize: lets a test model an index as another program might have written it
   (--loadIndex), e.g. with overlapping windows of one hash */
void orc_session_set_index(void* h, const orc_minmer* recs, int64_t n) {
  auto* S = (Session*)h;
  S->index.assign(recs, recs + n);
}
void orc_session_finalize(void* h) {
  auto* S = (Session*)h;
  build_lookup(*S);
  frequency_filter(*S);
  S->cutoffs.assign((size_t)(std::min<double>(S->sketchSize, SS_TABLE_MAX) + 1), 1);   /* computeMap.hpp:128 */
  S->refGroup.assign(S->meta.size(), 0);
  if (S->hg()) set_probs(*S);
  if (S->flags & ORC_SKIP_PREFIX) set_ref_groups(*S);
}
void orc_session_free(void* h) { delete (Session*)h; }

int64_t orc_session_index_size(void* h) { return (int64_t)((Session*)h)->index.size(); }
void orc_session_index_copy(void* h, orc_minmer* out) { auto& v = ((Session*)h)->index; std::copy(v.begin(), v.end(), out); }
int64_t orc_session_nkeys(void* h) { return (int64_t)((Session*)h)->lookup.size(); }
void orc_session_keys(void* h, uint64_t* keys, int64_t* counts) {
  size_t i = 0;
  for (auto& e : ((Session*)h)->lookup) { keys[i] = e.first; counts[i] = (int64_t)e.second.size(); i++; }
}
int64_t orc_session_lookup(void* h, uint64_t hash, orc_point* out, int64_t cap) {
  auto& m = ((Session*)h)->lookup;
  auto it = m.find(hash);
  if (it == m.end()) return -1;
  for (size_t i = 0; i < it->second.size() && (int64_t)i < cap; i++) out[i] = it->second[i];
  return (int64_t)it->second.size();
}
int64_t orc_session_npoints(void* h) { int64_t n = 0; for (auto& e : ((Session*)h)->lookup) n += (int64_t)e.second.size(); return n; }
/* whole lookup map, ascending key order: keys[i] owns pts[offsets[i] .. offsets[i+1]) */
void orc_session_export_lookup(void* h, uint64_t* keys, uint64_t* offsets, orc_point* pts) {
  size_t i = 0; uint64_t o = 0;
  for (auto& e : ((Session*)h)->lookup) {
    keys[i] = e.first; offsets[i] = o;
    for (auto& p : e.second) pts[o++] = p;
    i++;
  }
  offsets[i] = o;
}
int64_t orc_session_nfreq(void* h) { return (int64_t)((Session*)h)->frequent.size(); }
void orc_session_freq_list(void* h, uint64_t* out) { size_t i = 0; for (auto v : ((Session*)h)->frequent) out[i++] = v; }
int orc_session_is_freq(void* h, uint64_t hash) { return ((Session*)h)->frequent.count(hash) ? 1 : 0; }
int orc_session_freq_threshold(void* h) { return ((Session*)h)->freqThreshold; }
int orc_session_ncontigs(void* h) { return (int)((Session*)h)->meta.size(); }
int orc_session_contig_len(void* h, int i) { return ((Session*)h)->meta[i].len; }
int orc_session_ncutoffs(void* h) { return (int)((Session*)h)->cutoffs.size(); }
void orc_session_cutoffs(void* h, int* out) { auto& v = ((Session*)h)->cutoffs; std::copy(v.begin(), v.end(), out); }

int orc_session_map_fragment(void* h, const char* seq, int len, int fullLen, int seqCounter, const char* seqName,
                             orc_minmer* qsk, int qskCap, orc_point* pts, int ptsCap, orc_l1* l1, int l1Cap,
                             orc_l2* l2, int* l2cand, int l2Cap, orc_mapping* maps, int mapsCap,
                             int64_t* counts, double* kmerComplexity) {
  const Session& S = *(Session*)h;
  Frag Q; Q.seq.assign(seq, seq + len); Q.len = len; Q.fullLen = fullLen; Q.seqCounter = seqCounter; Q.name = seqName;
  Q.refGroup = get_ref_group(S, Q.name);
  std::vector<orc_point> pv; std::vector<orc_l1> l1v;
  const int minimumHits = l1_mapping(S, Q, pv, l1v);
  counts[0] = (int64_t)Q.sketch.size();
  for (size_t i = 0; i < Q.sketch.size() && (int)i < qskCap; i++) qsk[i] = Q.sketch[i];
  counts[1] = (int64_t)pv.size();
  for (size_t i = 0; i < pv.size() && (int)i < ptsCap; i++) pts[i] = pv[i];
  counts[2] = (int64_t)l1v.size();
  for (size_t i = 0; i < l1v.size() && (int)i < l1Cap; i++) l1[i] = l1v[i];
  int64_t nl2 = 0;
  for (size_t c = 0; c < l1v.size(); c++) {
    std::vector<orc_l2> loci;
    l2_regions(S, Q, l1v[c], loci);
    for (const auto& x : loci) { if (nl2 < l2Cap) { l2[nl2] = x; l2cand[nl2] = (int)c; } nl2++; }
  }
  counts[3] = nl2;
  counts[5] = Q.sketchSize > 0 ? minimumHits : 0;
  counts[6] = Q.sketchSize;
  counts[7] = Q.rawSketchSize;
  *kmerComplexity = Q.kmerComplexity;

  Frag Q2; Q2.seq.assign(seq, seq + len); Q2.len = len; Q2.fullLen = fullLen; Q2.seqCounter = seqCounter; Q2.name = seqName;
  Q2.refGroup = Q.refGroup;
  std::vector<orc_point> pv2; std::vector<orc_l1> l1v2; std::vector<orc_mapping> mv;
  map_fragment(S, Q2, pv2, l1v2, mv);
  counts[4] = (int64_t)mv.size();
  for (size_t i = 0; i < mv.size() && (int)i < mapsCap; i++) maps[i] = mv[i];
  return 0;
}

/* computeL1CandidateRegions (l1_candidates above) on a point list the caller made, sorted by (seqId, pos, side): lets a test check that a
 * transformation of the list (the device's pre-filter of points that cannot reach minimumHits, mm_map.hip k_filter_points) leaves the
 * candidates as they are.  Uses the session's segLength, sketchSize, HG flag and cut-off table; fragLen <= segLength (windowLen == 0). */
int orc_session_l1_from_points(void* h, const orc_point* pts, int64_t n, int qSketchSize, int fragLen, int minimumHits, orc_l1* out, int cap) {
  const Session& S = *(const Session*)h;
  Frag Q; Q.len = fragLen; Q.fullLen = fragLen; Q.seqCounter = 0; Q.refGroup = -1; Q.sketchSize = qSketchSize; Q.rawSketchSize = qSketchSize;
  std::vector<orc_l1> l1;
  if (n > 0) l1_candidates(S, Q, pts, pts + n, minimumHits, l1);
  for (size_t i = 0; i < l1.size() && (int)i < cap; i++) out[i] = l1[i];
  return (int)l1.size();
}

int orc_session_map_read(void*, const char*, int, int, const char*, orc_mapping*, int) { return -1; /* widened later */ }

}  // extern "C"

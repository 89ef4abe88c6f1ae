// tests/hostlogic/hostlogic.cpp -- TEST HARNESS: a C entry point around skch::MapPost (mashmap_amd/host/skch_map_post.hpp), the
// device-independent half of the host side, so that it can be driven on a machine without a GPU: the integers normally produced
// by the kernels are supplied by the caller (the tests take them from the real reference's own L1/L2 stages).
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../../mashmap_amd/host/skch_map_post.hpp"
#include "../../mashmap_amd/csrc/mm_select_core.h"

extern "C" {

struct hl_mapping {      // same fields as oracle.h's orc_mapping
  int32_t queryLen, refStartPos, refEndPos, queryStartPos, queryEndPos, refSeqId, querySeqId, blockLength;
  float nucIdentity, nucIdentityUpperBound;
  int32_t sketchSize, conservedSketches, strand, approxMatches;
  double kmerComplexity;
};

enum { HL_HG = 1, HL_SKIP_SELF = 2, HL_SKIP_PREFIX = 4, HL_LOWER_TRI = 8, HL_NOSPLIT = 16, HL_NOMERGE = 32, HL_DROP_LOW_ID = 64 };

int hl_map_read(int k, int segLength, int sketchSize, float pi, int filterMode, int flags, int numMappings,
                int nContigs, const char* const* names, const int32_t* lens, const int32_t* groups,
                const char* readName, int readLen, int seqCounter,
                int nFrags, const mm_fragment* frags, const mm_frag_stats* stats, int nL1, const mm_l1_candidate* l1, int nL2, const mm_l2_locus* l2,
                hl_mapping* out, int cap, char* paf, int pafCap) {
  skch::Parameters p;
  p.kmerSize = k; p.segLength = segLength; p.block_length = segLength; p.chain_gap = segLength; p.sketchSize = sketchSize;
  p.percentageIdentity = pi; p.filterMode = filterMode; p.stage1_topANI_filter = (flags & HL_HG) != 0;
  p.skip_self = (flags & HL_SKIP_SELF) != 0; p.skip_prefix = (flags & HL_SKIP_PREFIX) != 0; p.lower_triangular = (flags & HL_LOWER_TRI) != 0;
  p.split = !(flags & HL_NOSPLIT); p.mergeMappings = !(flags & HL_NOMERGE); p.keep_low_pct_id = !(flags & HL_DROP_LOW_ID);
  p.numMappingsForSegment = (uint32_t)numMappings; p.numMappingsForShortSequence = (uint32_t)numMappings;
  std::vector<skch::ContigInfo> meta;
  for (int i = 0; i < nContigs; i++) meta.push_back(skch::ContigInfo{names[i], lens[i]});
  std::vector<int> grp;
  if (groups) grp.assign(groups, groups + nContigs);
  skch::MapPost post(p, meta, grp);
  skch::DeviceResults D;
  D.frags.assign(frags, frags + nFrags); D.stats.assign(stats, stats + nFrags);
  D.l1.assign(l1, l1 + nL1); D.l2.assign(l2, l2 + nL2);
  D.fragBegin = {0, (size_t)nFrags};
  D.l1Begin.resize((size_t)nFrags + 1);
  { size_t o = 0; for (int f = 0; f < nFrags; f++) { D.l1Begin[f] = o; o += (size_t)stats[f].nL1; } D.l1Begin[nFrags] = o; }
  D.l2Begin.assign((size_t)nL1 + 1, (size_t)nL2);
  { size_t i = 0; for (int c = 0; c <= nL1; c++) { while (i < (size_t)nL2 && D.l2[i].cand < c) i++; D.l2Begin[c] = i; } }
  skch::MappingResultsVector_t res;
  post.mapModule(D, 0, readName, readLen, seqCounter, res);
  // The same read through the path the GPU run takes: doL2Mapping's walk in integers (mm_select_core.h: the code of k_l2_select,
  // with the host tables of mmhost::replayTables) -> mm_mapping records -> MapPost::mapModuleFromRecords.  Must give the same rows,
  // floats bit for bit; -2 reports a difference.
  {
    std::vector<uint8_t> accept; std::vector<int16_t> minIsz;
    mmhost::replayTables(sketchSize, k, pi, p.ANIDiff, p.keep_low_pct_id, 4, accept, minIsz);
    const size_t stride = (size_t)sketchSize + 1;
    std::vector<int64_t> l2First((size_t)nL1, 0); std::vector<int32_t> l2Num((size_t)nL1, 0);
    for (int c = 0; c < nL1; c++) { l2First[c] = (int64_t)D.l2Begin[c]; l2Num[c] = (int32_t)(D.l2Begin[c + 1] - D.l2Begin[c]); }
    std::vector<int32_t> heap((size_t)nL1 + 1);
    std::vector<int32_t> rg(grp.begin(), grp.end());
    if (rg.empty()) rg.assign((size_t)nContigs, 0);
    std::vector<mm_mapping> recs;
    for (int f = 0; f < nFrags; f++) {
      const int Qs = stats[f].sketchSize, nC = stats[f].nL1;
      if (Qs <= 0 || nC <= 0) continue;
      const size_t b = D.l1Begin[f];
      mm_mapping rec; std::memset(&rec, 0, sizeof rec);
      rec.querySeqId = seqCounter; rec.fragStart = frags[f].fragStart; rec.fragLen = frags[f].len; rec.sketchSize = Qs;
      rec.rawSketchSize = stats[f].rawSketchSize; rec.maxHash = stats[f].maxHash;
      mm_select_fragment(nC, D.l1.data() + b, heap.data(), l2First.data() + b, l2Num.data() + b, D.l2.data(), rg.data(), p.skip_prefix ? 1 : 0,
                         p.stage1_topANI_filter ? 1 : 0, Qs, accept.data() + (size_t)Qs * stride, minIsz.data() + (size_t)Qs * stride,
                         [&](const mm_l2_locus& L) { rec.refSeqId = L.seqId; rec.refStartPos = L.meanOptimalPos; rec.conservedSketches = L.sharedSketchSize; rec.strand = L.strand; recs.push_back(rec); });
    }
    skch::MappingResultsVector_t res2;
    if (readLen >= k) post.mapModuleFromRecords(recs.data(), recs.data() + recs.size(), readLen, res2);
    if (res2.size() != res.size()) return -2;
    for (size_t i = 0; i < res.size(); i++) {
      const auto& a = res[i]; const auto& b2 = res2[i];
      if (a.queryLen != b2.queryLen || a.refStartPos != b2.refStartPos || a.refEndPos != b2.refEndPos || a.queryStartPos != b2.queryStartPos ||
          a.queryEndPos != b2.queryEndPos || a.refSeqId != b2.refSeqId || a.querySeqId != b2.querySeqId || a.blockLength != b2.blockLength ||
          a.sketchSize != b2.sketchSize || a.conservedSketches != b2.conservedSketches || a.strand != b2.strand || a.approxMatches != b2.approxMatches ||
          a.n_merged != b2.n_merged || std::memcmp(&a.nucIdentity, &b2.nucIdentity, 4) || std::memcmp(&a.nucIdentityUpperBound, &b2.nucIdentityUpperBound, 4) ||
          a.kmerComplexity != b2.kmerComplexity) return -2;
    }
  }
  std::ostringstream os;
  post.reportReadMappings(res, readName, os);
  const std::string txt = os.str();
  if (paf && pafCap > 0) { std::strncpy(paf, txt.c_str(), (size_t)pafCap - 1); paf[pafCap - 1] = 0; }
  for (size_t i = 0; i < res.size() && (int)i < cap; i++) {
    const auto& e = res[i];
    out[i] = hl_mapping{e.queryLen, e.refStartPos, e.refEndPos, e.queryStartPos, e.queryEndPos, e.refSeqId, e.querySeqId, e.blockLength,
                        e.nucIdentity, e.nucIdentityUpperBound, e.sketchSize, e.conservedSketches, (int32_t)e.strand, e.approxMatches,
                        (double)e.kmerComplexity};
  }
  return (int)res.size();
}

// PAF / legacy text: MapPost::appendReadMappings (std::to_chars, what the pipeline uses) against the literal stream form
// (reportReadMappingsStream = computeMap.hpp:1758-1806) on n random mappings in every output mode.  Returns the number of modes x
// batches whose bytes differ.
int hl_paf_formatters_agree(int n, unsigned long long seed) {
  auto rnd = [&]() { seed += 0x9E3779B97F4A7C15ull; unsigned long long z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
  std::vector<skch::ContigInfo> meta{{"chr1", 248956422}, {"contig_with_a_long_name|x", 57}, {"c", 2147483647}};
  std::vector<skch::ContigInfo> qmeta{{"readA", 10000}, {"read/B 2", 123456}};
  int bad = 0;
  for (int mode = 0; mode < 16; mode++) {
    skch::Parameters p;
    p.legacy_output = (mode & 1) != 0; p.report_ANI_percentage = (mode & 2) != 0; p.mergeMappings = (mode & 4) == 0;
    p.filterMode = (mode & 8) ? skch::filter::ONETOONE : skch::filter::MAP;
    skch::MapPost post(p, meta, std::vector<int>());
    post.qmetadata = &qmeta;
    skch::MappingResultsVector_t v((size_t)n);
    for (auto& e : v) {
      e.queryLen = (skch::offset_t)(rnd() % 3000000); e.queryStartPos = (skch::offset_t)(rnd() % 100000); e.queryEndPos = e.queryStartPos + (skch::offset_t)(rnd() % 100000);
      e.refStartPos = (skch::offset_t)(rnd() % 2000000000); e.refEndPos = e.refStartPos + (skch::offset_t)(rnd() % 100000);
      e.refSeqId = (skch::seqno_t)(rnd() % 3); e.querySeqId = (skch::seqno_t)(rnd() % 2);
      e.strand = (rnd() & 1) ? skch::strnd::FWD : skch::strnd::REV;
      e.sketchSize = 1 + (int)(rnd() % 5000); e.conservedSketches = (int)(rnd() % (unsigned)(e.sketchSize + 1)); e.blockLength = (int)(rnd() % 1000000);
      const unsigned kind = (unsigned)(rnd() % 8);
      e.nucIdentity = kind == 0 ? 1.0f : kind == 1 ? 0.0f : kind == 2 ? (float)((rnd() % 1000) / 1000.0) : (float)((rnd() % 100000000) / 100000000.0);
      e.kmerComplexity = kind == 3 ? 1.0L : (long double)(rnd() % 100000) / (long double)(1 + rnd() % 100000);
    }
    std::ostringstream os; post.reportReadMappingsStream(v, "the/query name", os);
    std::string t; post.appendReadMappings(v, "the/query name", t);
    std::ostringstream os2; post.reportReadMappings(v, "the/query name", os2);
    if (os.str() != t || os2.str() != t) bad++;
  }
  return bad;
}

// The integer tables of mm_stats.hpp against their literal forms (round 4 shortened two loops of them without changing a bit): the
// binomial tail summed over ALL its terms (what gsl_cdf_binomial_Q's stand-in did until then), md_lower_bound on that sum, and the L1
// cut-off counted up as the reference does (computeMap.hpp:1196-1200).  Returns the number of entries that differ.
namespace lit {
inline double tailFull(unsigned k, double p, unsigned n) {
  if (k >= n) return 0.0;
  if (p <= 0.0) return 0.0;
  if (p >= 1.0) return 1.0;
  const double lp = std::log(p), lq = std::log1p(-p), mean = (double)n * p;
  double acc = 0.0;
  if ((double)k + 1.0 >= mean) { for (unsigned i = k + 1; i <= n; i++) acc += std::exp(mmhost::Stat::lnChoose(n, i) + i * lp + (double)(n - i) * lq); return acc > 1.0 ? 1.0 : acc; }
  for (unsigned i = 0; i <= k; i++) acc += std::exp(mmhost::Stat::lnChoose(n, i) + i * lp + (double)(n - i) * lq);
  return acc > 1.0 ? 0.0 : 1.0 - acc;
}
inline float mdLowerBound(float d, int s, int k, float ci) {
  using namespace mmhost::Stat;
  float q2 = (1.0 - ci) / 2;
  int x = std::max(int(std::ceil(s * md2j(d, k))), 1);
  while (x <= s) { double c = tailFull(x - 1, md2j(d, k), s); if (c < q2) { x--; break; } x++; }
  return j2md(float(x) / s, k);
}
}  // namespace lit
int hl_tables_vs_literal(int sketchSize, int k, float pi, int keepLow) {
  using namespace mmhost;
  std::vector<uint8_t> accept; std::vector<int16_t> minIsz;
  replayTables(sketchSize, k, pi, fixed::ANIDiff, keepLow != 0, 4, accept, minIsz);
  const size_t stride = (size_t)sketchSize + 1;
  int bad = 0;
  for (int Qs = 1; Qs <= sketchSize; Qs++) {
    for (int shared = 0; shared <= Qs; shared++) {
      const float md = Stat::j2md(1.0 * shared / Qs, k);
      bool ok = (1 - md) >= pi;
      if (!ok && keepLow) ok = (1 - lit::mdLowerBound(md, Qs, k, fixed::confidence_interval)) >= pi;
      if (accept[(size_t)Qs * stride + shared] != (ok ? 1 : 0)) bad++;
    }
    for (int best = 0; best <= Qs; best++) {
      const double cutoff_ani = std::max(0.0, double((1 - Stat::j2md((double)best / Qs, k)) - fixed::ANIDiff));
      const double cutoff_j = Stat::md2j(1 - cutoff_ani, k);
      int isz = 0; while (isz <= Qs && double(isz) / Qs < cutoff_j) isz++;
      if (minIsz[(size_t)Qs * stride + best] != (int16_t)isz) bad++;
    }
  }
  // the minimum-hits table walks md_lower_bound downwards from the first estimate (map_stats.hpp:144)
  const std::vector<int32_t> mh = minHitsTable(sketchSize, k, pi);
  for (int q = 1; q <= sketchSize; q++) {
    const int first = Stat::estimateMinimumHits(q, k, pi);
    int relaxed = first;
    for (int i = first; i >= 0; i--) { const float d = Stat::j2md((float)(1.0 * i / q), k); if (1.0 - lit::mdLowerBound(d, q, k, fixed::confidence_interval) >= pi) relaxed = i; else break; }
    if (mh[q] != relaxed) bad++;
  }
  return bad;
}
// the tail itself, bit for bit, on both of its branches; returns the number of (k, p, n) whose doubles differ
int hl_tail_vs_full(void) {
  int bad = 0;
  for (unsigned n : {1u, 10u, 130u, 1000u, 4000u, 9998u}) for (double p : {1e-6, 0.001, 0.02, 0.3, 0.5, 0.7, 0.999}) for (unsigned k = 0; k < n; k += (n / 61 + 1)) {
    const double a = mmhost::Stat::binomialUpperTail(k, p, n), b = lit::tailFull(k, p, n);
    if (std::memcmp(&a, &b, sizeof a) != 0) bad++;
  }
  return bad;
}

}  // extern "C"

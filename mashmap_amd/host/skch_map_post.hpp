// mashmap_amd/host/skch_map_post.hpp -- the host half of skch::Map that needs no device: everything from the integers of
// the hot path (fragment statistics, L1 candidates, L2 loci; in the layouts of include/mashmap_hip.h) to reported mappings.
//
//   doL2Mapping replay   best-first / early-exit over the L1 candidates, identities   computeMap.hpp:1182-1267
//   mapModule tail       coordinates of split reads, mergeMappingsInRange :1580-1702, filterWeakMappings :423,
//                        filterByGroup :504 with the plane sweeps of filter.hpp:103-160 / :334-396, filterFalseHighIdentity :441,
//                        mappingBoundarySanityCheck :1714, sparsifyMappings :481
//   reportReadMappings   the PAF / legacy text                                         computeMap.hpp:1758-1806
//
// skch::Map (skch_map.hpp) feeds it from the GPU; tests/hostlogic feeds it from the reference's own L1/L2 output on the CPU and
// compares with the reference's mapModule.
#pragma once
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <ostream>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/mashmap_hip.h"
#include "mm_stats.hpp"
#include "skch_types.hpp"

namespace skch {

namespace hipdetail {

// union-find with the tie rules of src/common/dset64.hpp:93-125 (union by rank; equal ranks: the smaller id becomes the root)
struct DisjointSets {
  std::vector<uint32_t> parent, rnk;
  explicit DisjointSets(size_t n) : parent(n), rnk(n, 0) { std::iota(parent.begin(), parent.end(), 0u); }
  uint32_t find(uint32_t x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
  }
  void unite(uint32_t a, uint32_t b) {
    a = find(a); b = find(b);
    if (a == b) return;
    if (rnk[a] > rnk[b] || (rnk[a] == rnk[b] && a < b)) std::swap(a, b);     // a goes under b
    parent[a] = b;
    if (rnk[a] == rnk[b]) rnk[b]++;
  }
};

// Plane sweep "best mapping(s) for every position" shared by both axes (filter.hpp:36-160 and :238-396).
// Pos is the sweep coordinate, Better orders the sweep-line status (best first).
template <typename Pos, typename Better>
void planeSweepFilter(MappingResultsVector_t& v, std::vector<std::tuple<Pos, int, int>>& events, Better better, int secondaryToKeep,
                      bool countBeforeCompare) {
  for (auto& e : v) e.discard = 1;
  auto score = [&v](int x) { return (double)v[x].nucIdentity; };
  std::set<int, Better> status(better);
  std::sort(events.begin(), events.end());
  for (size_t i = 0; i < events.size();) {
    size_t j = i;
    while (j < events.size() && std::get<0>(events[j]) == std::get<0>(events[i])) j++;
    for (size_t e = i; e < j; e++) {
      if (std::get<1>(events[e]) == event::BEGIN) status.insert(std::get<2>(events[e]));
      else status.erase(std::get<2>(events[e]));
    }
    if (!status.empty()) {                       // markGood (filter.hpp:77-95 / :285-301)
      const int best = *status.begin();
      int kept = 0;
      for (int id : status) {
        const bool worseOrSeen = score(best) > score(id) || v[id].discard == 0;
        if (countBeforeCompare) { if (worseOrSeen && ++kept > secondaryToKeep) break; }
        else if (worseOrSeen && kept > secondaryToKeep) break;
        v[id].discard = 0;
        if (!countBeforeCompare) ++kept;
      }
    }
    i = j;
  }
  v.erase(std::remove_if(v.begin(), v.end(), [](const MappingResult& e) { return e.discard == 1; }), v.end());
}

}  // namespace hipdetail

// integers of one device pass over a batch of reads, with the per-read / per-fragment / per-candidate ranges
struct DeviceResults {
  std::vector<mm_fragment> frags;
  std::vector<mm_frag_stats> stats;
  std::vector<mm_l1_candidate> l1;
  std::vector<mm_l2_locus> l2;
  std::vector<size_t> fragBegin;               // per read: first fragment
  std::vector<size_t> l1Begin;                 // per fragment: first L1 candidate
  std::vector<size_t> l2Begin;                 // per L1 candidate: first L2 locus
};


class MapPost {
  const skch::Parameters& param;
  const std::vector<ContigInfo>& metadata;         // reference contigs (Sketch::metadata)
  std::vector<int> refIdGroup;
  // nucIdentityUpperBound depends only on (sharedSketchSize, Q.sketchSize); md_lower_bound is a CDF search, so the few hundred
  // pairs that occur are computed once (a benign race: two threads may compute the same value)
  mutable std::vector<std::atomic<uint32_t>> ubCache;
  float identityUpperBound(float mash_dist, int shared, int Qs) const {
    const size_t at = (size_t)Qs * (size_t)(param.sketchSize + 1) + (size_t)shared;
    uint32_t bits = ubCache[at].load(std::memory_order_relaxed);
    if (bits == 0xFFFFFFFFu) {
      const float v = 1 - mmhost::Stat::md_lower_bound(mash_dist, Qs, param.kmerSize, skch::fixed::confidence_interval);
      std::memcpy(&bits, &v, 4);
      ubCache[at].store(bits, std::memory_order_relaxed);
    }
    float out; std::memcpy(&out, &bits, 4);
    return out;
  }


 public:
  const std::vector<ContigInfo>* qmetadata = nullptr;   // query names, only read when reporting one-to-one mappings

  MapPost(const skch::Parameters& p, const std::vector<ContigInfo>& refMetadata, const std::vector<int>& groups)
      : param(p), metadata(refMetadata), refIdGroup(groups), ubCache((size_t)(p.sketchSize + 1) * (size_t)(p.sketchSize + 1)) {
    if (refIdGroup.size() != metadata.size()) refIdGroup.assign(metadata.size(), 0);
    for (auto& e : ubCache) e.store(0xFFFFFFFFu, std::memory_order_relaxed);      // a NaN pattern no identity can have
  }
  const std::vector<int>& groups() const { return refIdGroup; }

  // ------------------------------------------------------------------------------------------------------------------
  // doL2Mapping (:1182-1267) for the candidates [b, e) of one fragment, on the device's integers
  void doL2MappingReplay(const DeviceResults& D, size_t f, std::vector<size_t>& cands, offset_t Qlen, seqno_t seqCounter, float kmerComplexity,
                         MappingResultsVector_t& l2Mappings) const {
    const int Qs = D.stats[f].sketchSize;
    auto cmp = [&D](size_t a, size_t b) { return D.l1[a].intersectionSize < D.l1[b].intersectionSize; };   // L1_locus_intersection_cmp (:71)
    if (param.stage1_topANI_filter) std::make_heap(cands.begin(), cands.end(), cmp);
    double bestJaccardNumerator = 0;
    size_t endIdx = cands.size(), it = 0;
    while (it != endIdx) {
      const size_t c = cands[it];
      if (param.stage1_topANI_filter) {
        const double cutoff_ani = std::max(0.0, double((1 - mmhost::Stat::j2md(bestJaccardNumerator / Qs, param.kmerSize)) - param.ANIDiff));
        const double cutoff_j = mmhost::Stat::md2j(1 - cutoff_ani, param.kmerSize);
        if (double(D.l1[c].intersectionSize) / Qs < cutoff_j) break;
      }
      for (size_t i = D.l2Begin[c]; i < D.l2Begin[c + 1]; i++) {
        const mm_l2_locus& l2 = D.l2[i];
        const float mash_dist = mmhost::Stat::j2md(1.0 * l2.sharedSketchSize / Qs, param.kmerSize);
        const float nucIdentity = (1 - mash_dist);
        const float nucIdentityUpperBound = identityUpperBound(mash_dist, l2.sharedSketchSize, Qs);
        if ((param.keep_low_pct_id && nucIdentityUpperBound >= param.percentageIdentity) || nucIdentity >= param.percentageIdentity) {
          bestJaccardNumerator = std::max<double>(bestJaccardNumerator, l2.sharedSketchSize);
          MappingResult res{};                 // n_merged / splitMappingId / discard are indeterminate in the reference (:1227); zero is what its
                                               // binary observably has there (a lone segment mapping of a longer read is dropped by filterWeakMappings)
          res.queryLen = Qlen;
          res.refStartPos = l2.meanOptimalPos;
          res.refEndPos = l2.meanOptimalPos + Qlen;
          res.queryStartPos = 0;
          res.queryEndPos = Qlen;
          res.refSeqId = l2.seqId;
          res.querySeqId = seqCounter;
          res.nucIdentity = nucIdentity;
          res.nucIdentityUpperBound = nucIdentityUpperBound;
          res.sketchSize = Qs;
          res.conservedSketches = l2.sharedSketchSize;
          res.blockLength = std::max(res.refEndPos - res.refStartPos, res.queryEndPos - res.queryStartPos);
          res.approxMatches = std::round(res.nucIdentity * res.blockLength / 100.0);
          res.strand = (strand_t)l2.strand;
          res.kmerComplexity = kmerComplexity;
          l2Mappings.push_back(res);
        }
      }
      if (param.stage1_topANI_filter) { std::pop_heap(cands.begin(), cands.begin() + endIdx, cmp); endIdx--; }
      else it++;
    }
  }

  // mapSingleQueryFrag (:756-815) minus the device part
  void fragmentMappings(const DeviceResults& D, size_t f, offset_t Qlen, seqno_t seqCounter, MappingResultsVector_t& l2Mappings) const {
    const mm_frag_stats& st = D.stats[f];
    if (st.sketchSize == 0 || st.rawSketchSize == 0) return;
    // getSeedHits (:830-831): long double ratio -> double -> float
    const double max_hash_01 = (long double)(st.maxHash) / std::numeric_limits<hash_t>::max();
    const float kmerComplexity = (double(st.rawSketchSize) / max_hash_01) / ((Qlen - param.kmerSize + 1) * 2);
    if (kmerComplexity < param.kmerComplexityThreshold) return;       // :1137
    const size_t b = D.l1Begin[f], e = D.l1Begin[f + 1];
    std::vector<size_t> cands;
    size_t gb = b;
    while (gb < e) {
      size_t ge = e;
      if (param.skip_prefix) { const int g = refIdGroup[D.l1[gb].seqId]; ge = gb; while (ge < e && refIdGroup[D.l1[ge].seqId] == g) ge++; }
      cands.resize(ge - gb);
      std::iota(cands.begin(), cands.end(), gb);
      doL2MappingReplay(D, f, cands, Qlen, seqCounter, kmerComplexity, l2Mappings);
      gb = ge;
    }
    std::sort(l2Mappings.begin(), l2Mappings.end(), [](const MappingResult& a, const MappingResult& b2) {
      return std::tie(a.refSeqId, a.refStartPos) < std::tie(b2.refSeqId, b2.refStartPos); });
  }

  // mapModule (:570-714) for read r of the batch
  void mapModule(const DeviceResults& D, size_t r, const std::string& name, offset_t len, seqno_t seqCounter, MappingResultsVector_t& out) const {
    (void)name;
    MappingResultsVector_t unfiltered, l2Mappings;
    bool split_mapping = true;
    const size_t fb = D.fragBegin[r], fe = D.fragBegin[r + 1];
    if (!param.split || len <= param.segLength) {
      if (fb < fe) fragmentMappings(D, fb, len, seqCounter, l2Mappings);
      unfiltered.insert(unfiltered.end(), l2Mappings.begin(), l2Mappings.end());
      split_mapping = false;
    } else {
      for (size_t f = fb; f < fe; f++) {
        l2Mappings.clear();
        fragmentMappings(D, f, D.frags[f].len, seqCounter, l2Mappings);
        for (auto& e : l2Mappings) { e.queryLen = len; e.queryStartPos = D.frags[f].fragStart; e.queryEndPos = D.frags[f].fragStart + D.frags[f].len; }
        unfiltered.insert(unfiltered.end(), l2Mappings.begin(), l2Mappings.end());
      }
    }
    finishRead(unfiltered, len, split_mapping, out);
  }

  // the tail of mapModule (:679-714): chaining, weak-chain filter, plane-sweep filter, sanity checks
  void finishRead(MappingResultsVector_t& unfiltered, offset_t len, bool split_mapping, MappingResultsVector_t& out) const {
    const int n_mappings = (int)(len < param.segLength ? param.numMappingsForShortSequence : param.numMappingsForSegment) - 1;
    if (split_mapping && param.mergeMappings) {
      mergeMappingsInRange(unfiltered, param.chain_gap);
      filterWeakMappings(unfiltered, (int64_t)std::floor(param.block_length / param.segLength));
    }
    if (param.filterMode == filter::MAP || param.filterMode == filter::ONETOONE) {
      MappingResultsVector_t tmp;
      filterByGroup(unfiltered, tmp, n_mappings, false);
      unfiltered = std::move(tmp);
    }
    out.swap(unfiltered);
    if (param.filterLengthMismatches) filterFalseHighIdentity(out);
    mappingBoundarySanityCheck(len, out);
    sparsifyMappings(out);
  }

  // identity of a locus sharing `shared` of Qs sketch elements (:1211-1213); a function of two small integers, cached like the bound
  float identityOf(int shared, int Qs, float* mashDist) const {
    const float mash_dist = mmhost::Stat::j2md(1.0 * shared / Qs, param.kmerSize);
    *mashDist = mash_dist;
    return 1 - mash_dist;
  }

  // mapModule (:570-714) for one read from the device's candidate mappings [b, e) (include/mashmap_hip.h: mm_mapping): the read's
  // records, fragment-major, inside a fragment in doL2Mapping's push order -- i.e. l2Mappings of mapSingleQueryFrag before its sort
  void mapModuleFromRecords(const mm_mapping* b, const mm_mapping* e, offset_t len, MappingResultsVector_t& out) const {
    MappingResultsVector_t unfiltered, l2Mappings;
    const bool split_mapping = param.split && len > param.segLength;
    for (const mm_mapping* p = b; p != e;) {
      const mm_mapping* q = p;
      while (q != e && q->fragStart == p->fragStart) q++;
      const offset_t Qlen = p->fragLen;
      // getSeedHits (:830-831): long double ratio -> double -> float
      const double max_hash_01 = (long double)(p->maxHash) / std::numeric_limits<hash_t>::max();
      const float kmerComplexity = (double(p->rawSketchSize) / max_hash_01) / ((Qlen - param.kmerSize + 1) * 2);
      if (!(kmerComplexity < param.kmerComplexityThreshold)) {          // :1137
        l2Mappings.clear();
        for (const mm_mapping* m = p; m != q; ++m) {
          float mash_dist;
          MappingResult res{};                 // see doL2MappingReplay for the zero-initialised tail
          res.nucIdentity = identityOf(m->conservedSketches, m->sketchSize, &mash_dist);
          res.nucIdentityUpperBound = identityUpperBound(mash_dist, m->conservedSketches, m->sketchSize);
          res.queryLen = Qlen;
          res.refStartPos = m->refStartPos;
          res.refEndPos = m->refStartPos + Qlen;
          res.queryStartPos = 0;
          res.queryEndPos = Qlen;
          res.refSeqId = m->refSeqId;
          res.querySeqId = m->querySeqId;
          res.sketchSize = m->sketchSize;
          res.conservedSketches = m->conservedSketches;
          res.blockLength = std::max(res.refEndPos - res.refStartPos, res.queryEndPos - res.queryStartPos);
          res.approxMatches = std::round(res.nucIdentity * res.blockLength / 100.0);
          res.strand = (strand_t)m->strand;
          res.kmerComplexity = kmerComplexity;
          l2Mappings.push_back(res);
        }
        std::sort(l2Mappings.begin(), l2Mappings.end(), [](const MappingResult& a, const MappingResult& b2) {
          return std::tie(a.refSeqId, a.refStartPos) < std::tie(b2.refSeqId, b2.refStartPos); });       // :799-800
        if (split_mapping)
          for (auto& x : l2Mappings) { x.queryLen = len; x.queryStartPos = p->fragStart; x.queryEndPos = p->fragStart + p->fragLen; }   // :632-636
        unfiltered.insert(unfiltered.end(), l2Mappings.begin(), l2Mappings.end());
      }
      p = q;
    }
    finishRead(unfiltered, len, split_mapping, out);
  }

  // ------------------------------------------------------------------------------------------------------------------
  void mergeMappingsInRange(MappingResultsVector_t& v, int max_dist) const {      // :1580-1702
    if (v.size() < 2) return;
    std::sort(v.begin(), v.end(), [](const MappingResult& a, const MappingResult& b) {
      return std::tie(a.refSeqId, a.refStartPos, a.queryStartPos) < std::tie(b.refSeqId, b.refStartPos, b.queryStartPos); });
    for (size_t i = 0; i < v.size(); i++) { v[i].splitMappingId = (offset_t)i; v[i].discard = 0; }
    hipdetail::DisjointSets sets(v.size());
    std::vector<std::pair<double, uint64_t>> distances;
    for (size_t i = 0; i < v.size(); i++) {
      const MappingResult& a = v[i];
      distances.clear();
      for (size_t j = i + 1; j < v.size(); j++) {
        const MappingResult& b = v[j];
        if (b.refSeqId != a.refSeqId || b.refStartPos > a.refEndPos + max_dist) break;
        if (b.strand != a.strand) continue;
        const int ref_dist = b.refStartPos - a.refEndPos;
        int query_dist = 0;
        double dist = std::numeric_limits<double>::max(), score = std::numeric_limits<double>::max();
        if (a.strand == strnd::FWD && a.queryStartPos <= b.queryStartPos) {
          query_dist = b.queryStartPos - a.queryEndPos;
          dist = std::sqrt(std::pow(query_dist, 2) + std::pow(ref_dist, 2));
          score = std::pow(query_dist - ref_dist, 2);
        } else if (a.strand != strnd::FWD && a.queryEndPos >= b.queryEndPos) {
          query_dist = a.queryStartPos - b.queryEndPos;
          dist = std::sqrt(std::pow(query_dist, 2) + std::pow(ref_dist, 2));
          score = std::pow(query_dist - ref_dist, 2);
        }
        if (dist < max_dist) distances.push_back(std::make_pair(dist + score, (uint64_t)b.splitMappingId));
      }
      if (!distances.empty()) {
        std::sort(distances.begin(), distances.end());
        sets.unite((uint32_t)a.splitMappingId, (uint32_t)distances.front().second);
      }
    }
    for (auto& e : v) e.splitMappingId = (offset_t)sets.find((uint32_t)e.splitMappingId);
    std::sort(v.begin(), v.end(), [](const MappingResult& a, const MappingResult& b) { return a.splitMappingId < b.splitMappingId; });
    for (auto it = v.begin(); it != v.end();) {
      auto it_end = std::find_if(it, v.end(), [&](const MappingResult& e) { return e.splitMappingId != it->splitMappingId; });
      for (auto m = it; m != it_end; ++m) {
        it->queryStartPos = std::min(it->queryStartPos, m->queryStartPos);
        it->refStartPos = std::min(it->refStartPos, m->refStartPos);
        it->queryEndPos = std::max(it->queryEndPos, m->queryEndPos);
        it->refEndPos = std::max(it->refEndPos, m->refEndPos);
        it->blockLength = std::max(it->refEndPos - it->refStartPos, it->queryEndPos - it->queryStartPos);
        it->approxMatches = std::round(it->nucIdentity * it->blockLength / 100.0);
      }
      it->n_merged = (int)std::distance(it, it_end);
      it->nucIdentity = (std::accumulate(it, it_end, 0.0, [](double x, MappingResult& e) { return x + e.nucIdentity; })) / it->n_merged;
      it->kmerComplexity = (std::accumulate(it, it_end, 0.0, [](double x, MappingResult& e) { return x + e.kmerComplexity; })) / it->n_merged;
      for (auto m = std::next(it); m != it_end; ++m) m->discard = 1;
      it = it_end;
    }
    v.erase(std::remove_if(v.begin(), v.end(), [](const MappingResult& e) { return e.discard == 1; }), v.end());
  }

  void filterWeakMappings(MappingResultsVector_t& v, int64_t min_count) const {    // :423-432
    v.erase(std::remove_if(v.begin(), v.end(), [&](const MappingResult& e) { return e.queryLen > e.blockLength && e.n_merged < min_count; }), v.end());
  }

  void filterFalseHighIdentity(MappingResultsVector_t& v) const {                  // :441-454
    v.erase(std::remove_if(v.begin(), v.end(), [&](const MappingResult& e) {
      const int64_t q_l = (int64_t)e.queryEndPos - (int64_t)e.queryStartPos;
      const int64_t r_l = (int64_t)e.refEndPos + 1 - (int64_t)e.refStartPos;
      const uint64_t delta = std::abs(r_l - q_l);
      const float len_id_bound = (1.0 - (float)delta / (float)q_l);
      return len_id_bound < std::min(0.7, std::pow(param.percentageIdentity, 3));
    }), v.end());
  }

  void sparsifyMappings(MappingResultsVector_t& v) const {                         // :481-492
    if (param.sparsity_hash_threshold < std::numeric_limits<uint64_t>::max())
      v.erase(std::remove_if(v.begin(), v.end(), [&](MappingResult& e) { return e.hash() > param.sparsity_hash_threshold; }), v.end());
  }

  void mappingBoundarySanityCheck(offset_t inputLen, MappingResultsVector_t& v) const {   // :1714-1750
    for (auto& e : v) {
      const offset_t rlen = metadata[e.refSeqId].len;
      if (e.refStartPos < 0) e.refStartPos = 0;
      if (e.refStartPos >= rlen) e.refStartPos = rlen - 1;
      if (e.refEndPos < e.refStartPos) e.refEndPos = e.refStartPos;
      if (e.refEndPos >= rlen) e.refEndPos = rlen - 1;
      if (e.queryStartPos < 0) e.queryStartPos = 0;
      if (e.queryStartPos >= inputLen) e.queryStartPos = inputLen;
      if (e.queryEndPos < e.queryStartPos) e.queryEndPos = e.queryStartPos;
      if (e.queryEndPos >= inputLen) e.queryEndPos = inputLen;
    }
  }

  // filter.hpp:103-160 (query axis) and :334-396 (reference axis)
  void filterQueryAxis(MappingResultsVector_t& v, int secondaryToKeep) const {
    if (v.size() <= 1) return;
    auto better = [&v](int x, int y) {
      const double xs = v[x].nucIdentity, ys = v[y].nucIdentity;
      return std::tie(xs, v[x].queryStartPos, v[x].refSeqId) > std::tie(ys, v[y].queryStartPos, v[y].refSeqId);
    };
    std::vector<std::tuple<offset_t, int, int>> events;
    events.reserve(2 * v.size());
    for (int i = 0; i < (int)v.size(); i++) { events.emplace_back(v[i].queryStartPos, (int)event::BEGIN, i); events.emplace_back(v[i].queryEndPos, (int)event::END, i); }
    hipdetail::planeSweepFilter<offset_t>(v, events, better, secondaryToKeep, false);
  }
  void filterRefAxis(MappingResultsVector_t& v, int secondaryToKeep) const {
    if (v.size() <= 1) return;
    auto better = [&v](int x, int y) {
      const double xs = v[x].nucIdentity, ys = v[y].nucIdentity;
      return std::tie(xs, v[x].refStartPos) > std::tie(ys, v[y].refStartPos);
    };
    typedef std::pair<seqno_t, offset_t> RefPos;
    std::vector<std::tuple<RefPos, int, int>> events;
    events.reserve(2 * v.size());
    for (int i = 0; i < (int)v.size(); i++) {
      events.emplace_back(RefPos(v[i].refSeqId, v[i].refStartPos), (int)event::BEGIN, i);
      RefPos endp(v[i].refSeqId, v[i].refEndPos);                        // refPosDoPlusOne (:309-322)
      if (endp.second == metadata[endp.first].len - 1) { endp.first += 1; endp.second = 0; } else endp.second += 1;
      events.emplace_back(endp, (int)event::END, i);
    }
    hipdetail::planeSweepFilter<RefPos>(v, events, better, secondaryToKeep, true);
  }

  void filterByGroup(MappingResultsVector_t& unfiltered, MappingResultsVector_t& filtered, int n_mappings, bool filter_ref) const {   // :504-561
    filtered.reserve(filtered.size() + unfiltered.size());
    std::sort(unfiltered.begin(), unfiltered.end(), [](const MappingResult& a, const MappingResult& b) {
      return std::tie(a.refSeqId, a.refStartPos) < std::tie(b.refSeqId, b.refStartPos); });
    if (param.filterMode == filter::MAP || param.filterMode == filter::ONETOONE) {
      MappingResultsVector_t tmp;
      auto b = unfiltered.begin();
      while (b != unfiltered.end()) {
        auto e = unfiltered.end();
        if (param.skip_prefix) {
          const int g = refIdGroup[b->refSeqId];
          e = std::find_if_not(b, unfiltered.end(), [&](const MappingResult& m) { return g == refIdGroup[m.refSeqId]; });
        }
        tmp.assign(b, e);
        std::sort(tmp.begin(), tmp.end(), [](const MappingResult& a, const MappingResult& c) {
          return std::tie(a.queryStartPos, a.refSeqId, a.refStartPos) < std::tie(c.queryStartPos, c.refSeqId, c.refStartPos); });
        if (filter_ref) filterRefAxis(tmp, (uint16_t)n_mappings);
        else filterQueryAxis(tmp, (uint16_t)n_mappings);
        filtered.insert(filtered.end(), tmp.begin(), tmp.end());
        tmp.clear();
        b = e;
      }
    }
    std::sort(filtered.begin(), filtered.end(), [](const MappingResult& a, const MappingResult& b) {
      return std::tie(a.queryStartPos, a.refSeqId, a.refStartPos) < std::tie(b.queryStartPos, b.refSeqId, b.refStartPos); });
  }

  // PAF text (:1758-1806); the caller invokes processMappingResults afterwards, in output order
  // computeMap.hpp:1758-1806, literally: one insertion after the other into a stream.  This form is the CHECKER of appendReadMappings
  // below (tests/test_host_logic.py runs both on the same mappings in every output mode); the pipeline uses the other one.
  void reportReadMappingsStream(const MappingResultsVector_t& readMappings, const std::string& queryName, std::ostream& outstrm) const {
    for (auto& e : readMappings) {
      const float fakeMapQ = e.nucIdentity == 1 ? 255 : std::round(-10.0 * std::log10(1 - (e.nucIdentity)));
      const std::string sep = param.legacy_output ? " " : "\t";
      outstrm << (param.filterMode == filter::ONETOONE ? (*qmetadata)[e.querySeqId].name : queryName)
              << sep << e.queryLen << sep << e.queryStartPos << sep << e.queryEndPos - (param.legacy_output ? 1 : 0)
              << sep << (e.strand == strnd::FWD ? "+" : "-")
              << sep << metadata[e.refSeqId].name << sep << metadata[e.refSeqId].len
              << sep << e.refStartPos << sep << e.refEndPos - (param.legacy_output ? 1 : 0);
      if (!param.legacy_output) {
        outstrm << sep << e.conservedSketches << sep << e.blockLength << sep << fakeMapQ
                << sep << "id:f:" << (param.report_ANI_percentage ? 100.0 : 1.0) * e.nucIdentity
                << sep << "kc:f:" << e.kmerComplexity;
        if (!param.mergeMappings) outstrm << sep << "jc:f:" << float(e.conservedSketches) / e.sketchSize;
      } else {
        outstrm << sep << e.nucIdentity * 100.0;
      }
      outstrm << "\n";
    }
  }

  // The same text appended to a string, field by field with std::to_chars: a stream insertion of a float goes through the locale
  // machinery (~0.3 us each, three per line), and a million reads a second is what the post stage has to keep up with.  An unformatted
  // operator<< of a float / double / long double is printf's %g with precision 6, which is chars_format::general with precision 6.
  void appendReadMappings(const MappingResultsVector_t& readMappings, const std::string& queryName, std::string& out) const {
    char b[64];
    auto num = [&](long long v) { const auto r = std::to_chars(b, b + sizeof b, v); out.append(b, r.ptr); };
    auto flt = [&](auto v) { const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::general, 6); out.append(b, r.ptr); };
    const char sep = param.legacy_output ? ' ' : '\t';
    const int back = param.legacy_output ? 1 : 0;
    for (const auto& e : readMappings) {
      const float fakeMapQ = e.nucIdentity == 1 ? 255 : std::round(-10.0 * std::log10(1 - (e.nucIdentity)));
      out += param.filterMode == filter::ONETOONE ? (*qmetadata)[e.querySeqId].name : queryName;
      out += sep; num(e.queryLen); out += sep; num(e.queryStartPos); out += sep; num(e.queryEndPos - back);
      out += sep; out += e.strand == strnd::FWD ? '+' : '-';
      out += sep; out += metadata[e.refSeqId].name; out += sep; num(metadata[e.refSeqId].len);
      out += sep; num(e.refStartPos); out += sep; num(e.refEndPos - back);
      if (!param.legacy_output) {
        out += sep; num(e.conservedSketches); out += sep; num(e.blockLength); out += sep; flt(fakeMapQ);
        out += sep; out += "id:f:"; flt((param.report_ANI_percentage ? 100.0 : 1.0) * e.nucIdentity);
        out += sep; out += "kc:f:"; flt(e.kmerComplexity);
        if (!param.mergeMappings) { out += sep; out += "jc:f:"; flt(float(e.conservedSketches) / e.sketchSize); }
      } else {
        out += sep; flt(e.nucIdentity * 100.0);
      }
      out += '\n';
    }
  }

  // the reference's name and signature (Map::reportReadMappings): the text goes to a stream
  void reportReadMappings(MappingResultsVector_t& readMappings, const std::string& queryName, std::ostream& outstrm) const {
    std::string t;
    appendReadMappings(readMappings, queryName, t);
    outstrm.write(t.data(), (std::streamsize)t.size());
  }
};

}  // namespace skch

// mashmap_amd/host/skch_map.hpp -- skch::Map on top of the C ABI (include/mashmap_hip.h).
//
// Stands in for the reference's class of the same name (src/map/include/computeMap.hpp:53-1820): same constructor
// signature (maps every query and writes param.outFileName in the ctor, :123-141), same optional per-mapping callback
// (:100-101, called once per reported MappingResult, :1802), same PAF text (:1758-1806).
//
// What moved: sketchSequence, getSeedHits, getSeedIntervalPoints, computeL1CandidateRegions and
// computeL2MappedRegions (:818-1451) run on the GPU for a whole batch of reads at once (mm_map_fragments).  What stays
// here, on host threads, is everything that touches floating point or std:: algorithm order:
//   * the best-first / early-exit replay of doL2Mapping (:1182-1267) over the device's integer L2 loci,
//   * chaining (mergeMappingsInRange :1580-1702), filterWeakMappings :423, the plane-sweep filters
//     (filter.hpp:103-160 query axis, :334-396 reference axis), filterFalseHighIdentity :441,
//     mappingBoundarySanityCheck :1714, sparsifyMappings :481 and the PAF writer.
// The reference runs one pthread task per read (ThreadPool.hpp); here the reader accumulates reads into batches
// (MASHMAP_HIP_BATCH_MBP, default 512 Mbp), one batch is one device pass, and the per-read host work of a batch is
// spread over param.threads std::threads.  Output order == input order, as in the reference (ThreadPool.hpp:187-211).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <mutex>
#include <numeric>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/mashmap_hip.h"
#include "mm_stats.hpp"
#include "seq_reader.hpp"
#include "skch_sketch.hpp"
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

class Map {
 public:
  struct L1_candidateLocus_t { seqno_t seqId; offset_t rangeStartPos; offset_t rangeEndPos; int intersectionSize; };   // computeMap.hpp:58
  struct L2_mapLocus_t { seqno_t seqId; offset_t meanOptimalPos, optimalStart, optimalEnd; int sharedSketchSize; strand_t strand; };   // :76
  typedef std::function<void(const MappingResult&)> PostProcessResultsFn_t;

 private:
  const skch::Parameters& param;
  const skch::Sketch& refSketch;
  PostProcessResultsFn_t processMappingResults;
  std::vector<ContigInfo> qmetadata;             // only filled for one-to-one filtering (:105)
  std::vector<int> refIdGroup;
  std::unordered_map<std::string, int> refNameToId;
  mm_ctx* ctx;
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

  struct Batch {
    std::string bases;
    std::vector<int64_t> offs{0};
    std::vector<std::string> names;
    seqno_t firstSeqCounter = 0;
    void clear(seqno_t next) { bases.clear(); offs.assign(1, 0); names.clear(); firstSeqCounter = next; }
    size_t size() const { return names.size(); }
  };
  struct DeviceResults {
    std::vector<mm_fragment> frags;
    std::vector<mm_frag_stats> stats;
    std::vector<mm_l1_candidate> l1;
    std::vector<mm_l2_locus> l2;
    std::vector<size_t> fragBegin;               // per read: first fragment
    std::vector<size_t> l1Begin;                 // per fragment: first L1 candidate
    std::vector<size_t> l2Begin;                 // per L1 candidate: first L2 locus
  };

  [[noreturn]] void die(const char* what) const {
    std::cerr << "[mashmap_hip::skch::Map] ERROR: " << what << ": " << mm_last_error(ctx) << std::endl;
    exit(1);
  }

 public:
  Map(const skch::Parameters& p, const skch::Sketch& refsketch, PostProcessResultsFn_t f = nullptr)
      : param(p), refSketch(refsketch), processMappingResults(f), refIdGroup(refsketch.metadata.size(), 0), ctx(refsketch.ctx()),
        ubCache((size_t)(p.sketchSize + 1) * (size_t)(p.sketchSize + 1)) {
    for (auto& e : ubCache) e.store(0xFFFFFFFFu, std::memory_order_relaxed);      // a NaN pattern no identity can have
    if (p.skip_prefix) refIdGroup = refsketch.refGroups();
    for (size_t i = 0; i < refsketch.metadata.size(); i++) refNameToId.emplace(refsketch.metadata[i].name, (int)i);
    // integer tables the kernels consume: estimateMinimumHitsRelaxed per Q.sketchSize (:1144) and sketchCutoffs (:178-258)
    std::vector<int32_t> minHits((size_t)p.sketchSize + 1, 0);
    for (int q = 1; q <= p.sketchSize; q++)
      minHits[q] = mmhost::Stat::estimateMinimumHitsRelaxed(q, p.kmerSize, p.percentageIdentity, skch::fixed::confidence_interval);
    std::vector<int> cut = mmhost::sketchCutoffs(p.sketchSize, p.kmerSize, p.ANIDiff, p.ANIDiffConf, p.stage1_topANI_filter);
    std::vector<int32_t> cut32(cut.begin(), cut.end());
    if (mm_set_tables(ctx, minHits.data(), minHits.size(), cut32.data(), cut32.size()) != MM_OK) die("mm_set_tables");
    this->mapQuery();
  }

  static void insertL2ResultsToVec(MappingResultsVector_t& v, const MappingResult& reportedL2Result) { v.push_back(reportedL2Result); }

 private:
  std::string prefix(const std::string& s) const { return s.substr(0, s.find_last_of(param.prefix_delim)); }

  int getRefGroup(const std::string& seqName) const {      // :164-176
    const std::string qp = prefix(seqName);
    for (size_t i = 0; i < refSketch.metadata.size(); i++)
      if (qp == prefix(refSketch.metadata[i].name)) return refIdGroup[i];
    return -1;
  }

  // ------------------------------------------------------------------------------------------------------------------
  void mapQuery() {                                         // :263-420
    seqno_t totalReadsPickedForMapping = 0, totalReadsMapped = 0, seqCounter = 0;
    uint64_t totalBp = 0;
    std::ofstream outstrm(param.outFileName);
    MappingResultsVector_t allReadMappings;
    const char* be = getenv("MASHMAP_HIP_BATCH_MBP");
    const size_t batchBases = (size_t)((be ? atof(be) : 512.0) * 1e6);
    // reader thread: parses the query files into batches (at most two waiting) while this thread drives the device pass and
    // the host post-processing of the previous batch
    std::deque<Batch> ready; std::mutex mu; std::condition_variable cvFull, cvEmpty; bool readerDone = false;
    std::thread reader([&]() {
      Batch batch;
      auto flush = [&]() {
        if (batch.size() == 0) return;
        std::unique_lock<std::mutex> lk(mu);
        cvFull.wait(lk, [&] { return ready.size() < 2; });
        ready.emplace_back(std::move(batch));
        lk.unlock(); cvEmpty.notify_one();
        batch = Batch(); batch.clear(seqCounter);
      };
      for (const auto& fileName : param.querySequences) {
        mmhost::for_each_seq_in_file(fileName, {}, "", [&](const std::string& name, std::string& seq) {
          const offset_t len = (offset_t)seq.length();
          totalBp += seq.length();
          if (param.filterMode == filter::ONETOONE) qmetadata.push_back(ContigInfo{name, len});
          if (len < param.kmerSize) {
            std::cerr << std::endl << "WARNING, skch::Map::mapQuery, read " << name << " of " << len << "bp "
                      << " is not long enough for mapping at segment length " << param.segLength << std::endl;
          } else {
            totalReadsPickedForMapping++;
          }
          // short reads travel too (they yield no fragment) so that seqCounter == firstSeqCounter + index inside the batch
          batch.names.push_back(name);
          batch.bases.append(seq);
          batch.offs.push_back((int64_t)batch.bases.size());
          seqCounter++;
          if (batch.bases.size() >= batchBases) flush();
        });
      }
      flush();
      { std::lock_guard<std::mutex> lk(mu); readerDone = true; }
      cvEmpty.notify_one();
    });
    while (true) {
      Batch cur;
      {
        std::unique_lock<std::mutex> lk(mu);
        cvEmpty.wait(lk, [&] { return !ready.empty() || readerDone; });
        if (ready.empty()) break;
        cur = std::move(ready.front()); ready.pop_front();
      }
      cvFull.notify_one();
      processBatch(cur, allReadMappings, totalReadsMapped, outstrm);
    }
    reader.join();

    if (param.filterMode == filter::ONETOONE) {            // :358-406
      const int n_mappings = (int)param.numMappingsForSegment - 1;
      MappingResultsVector_t tmp, filtered;
      auto b = allReadMappings.begin();
      while (b != allReadMappings.end()) {
        auto e = allReadMappings.end();
        if (param.skip_prefix) {
          const int g = getRefGroup(qmetadata[b->querySeqId].name);
          e = std::find_if_not(b, allReadMappings.end(), [&](const MappingResult& m) { return g == getRefGroup(qmetadata[m.querySeqId].name); });
        }
        tmp.assign(b, e);
        filterByGroup(tmp, filtered, n_mappings, true);
        tmp.clear();
        b = e;
      }
      allReadMappings = std::move(filtered);
      std::sort(allReadMappings.begin(), allReadMappings.end(), [](const MappingResult& a, const MappingResult& b2) {
        return std::tie(a.querySeqId, a.queryStartPos, a.refSeqId, a.refStartPos) < std::tie(b2.querySeqId, b2.queryStartPos, b2.refSeqId, b2.refStartPos);
      });
      std::ostringstream os;
      reportReadMappings(allReadMappings, "", os);
      outstrm << os.str();
      if (processMappingResults) for (const auto& e : allReadMappings) processMappingResults(e);
    }
    std::cerr << "[mashmap::skch::Map::mapQuery] count of mapped reads = " << totalReadsMapped
              << ", reads qualified for mapping = " << totalReadsPickedForMapping << ", total input reads = " << seqCounter
              << ", total input bp = " << totalBp << std::endl;
  }

  // ------------------------------------------------------------------------------------------------------------------
  // one device pass + the per-read host post-processing of a batch
  void processBatch(const Batch& batch, MappingResultsVector_t& allReadMappings, seqno_t& totalReadsMapped, std::ofstream& outstrm) {
    const size_t nReads = batch.size();
    const bool timing = getenv("MASHMAP_HIP_TIMING") != nullptr;
    auto tick = skch::Time::now();
    auto lap = [&](const char* what) {
      if (!timing) return;
      const auto now = skch::Time::now();
      std::cerr << "[mashmap_hip::timing] " << what << ": " << std::chrono::duration<double>(now - tick).count() << " s" << std::endl;
      tick = now;
    };
    std::vector<int32_t> readGroup, readSelf;
    if (param.skip_prefix) { readGroup.resize(nReads); for (size_t r = 0; r < nReads; r++) readGroup[r] = getRefGroup(batch.names[r]); }
    if (param.skip_self) {
      readSelf.resize(nReads);
      for (size_t r = 0; r < nReads; r++) { auto it = refNameToId.find(batch.names[r]); readSelf[r] = it == refNameToId.end() ? -1 : it->second; }
    }
    if (mm_reads_upload(ctx, batch.bases.data(), batch.offs.data(), nReads, param.skip_prefix ? readGroup.data() : nullptr,
                        param.skip_self ? readSelf.data() : nullptr, batch.firstSeqCounter) != MM_OK) die("mm_reads_upload");
    lap("upload + pack");
    if (mm_map_fragments(ctx) != MM_OK) die("mm_map_fragments");
    lap("device pass");
    DeviceResults D;
    size_t n1 = 0, n2 = 0;
    if (mm_result_counts(ctx, &n1, &n2) != MM_OK) die("mm_result_counts");
    const size_t nF = mm_num_fragments(ctx);
    D.frags.resize(nF); D.stats.resize(nF); D.l1.resize(n1); D.l2.resize(n2);
    if (mm_fragments_download(ctx, D.frags.data()) != MM_OK) die("mm_fragments_download");
    if (mm_results_download(ctx, D.stats.data(), D.l1.data(), D.l2.data()) != MM_OK) die("mm_results_download");
    D.fragBegin.assign(nReads + 1, nF);
    { size_t f = 0; for (size_t r = 0; r <= nReads; r++) { while (f < nF && (size_t)D.frags[f].readId < r) f++; D.fragBegin[r] = f; } }
    D.l1Begin.resize(nF + 1);
    { size_t o = 0; for (size_t f = 0; f < nF; f++) { D.l1Begin[f] = o; o += (size_t)D.stats[f].nL1; } D.l1Begin[nF] = o; }
    D.l2Begin.assign(n1 + 1, n2);
    { size_t i = 0; for (size_t c = 0; c <= n1; c++) { while (i < n2 && (size_t)D.l2[i].cand < c) i++; D.l2Begin[c] = i; } }

    lap("download");
    std::vector<MappingResultsVector_t> perRead(nReads);
    std::vector<std::string> text(nReads);
    const bool reportNow = param.filterMode != filter::ONETOONE;
    const unsigned nThreads = (unsigned)std::max(1, param.threads);
    auto work = [&](unsigned t) {
      std::ostringstream os;
      for (size_t r = t; r < nReads; r += nThreads) {
        const offset_t len = (offset_t)(batch.offs[r + 1] - batch.offs[r]);
        if (len < param.kmerSize) continue;
        mapModule(D, r, batch.names[r], len, batch.firstSeqCounter + (seqno_t)r, perRead[r]);
        if (reportNow && !perRead[r].empty()) { os.str(std::string()); reportReadMappings(perRead[r], batch.names[r], os); text[r] = os.str(); }
      }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nThreads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();

    lap("host replay + chain + filter");
    for (size_t r = 0; r < nReads; r++) {                  // mapModuleHandleOutput (:724-752), input order
      if (!perRead[r].empty()) totalReadsMapped++;
      if (!reportNow) allReadMappings.insert(allReadMappings.end(), perRead[r].begin(), perRead[r].end());
      else {
        outstrm << text[r];
        if (processMappingResults) for (const auto& e : perRead[r]) processMappingResults(e);
      }
    }
  }

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
      const offset_t rlen = refSketch.metadata[e.refSeqId].len;
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
      if (endp.second == refSketch.metadata[endp.first].len - 1) { endp.first += 1; endp.second = 0; } else endp.second += 1;
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
  void reportReadMappings(MappingResultsVector_t& readMappings, const std::string& queryName, std::ostream& outstrm) const {
    for (auto& e : readMappings) {
      const float fakeMapQ = e.nucIdentity == 1 ? 255 : std::round(-10.0 * std::log10(1 - (e.nucIdentity)));
      const std::string sep = param.legacy_output ? " " : "\t";
      outstrm << (param.filterMode == filter::ONETOONE ? qmetadata[e.querySeqId].name : queryName)
              << sep << e.queryLen << sep << e.queryStartPos << sep << e.queryEndPos - (param.legacy_output ? 1 : 0)
              << sep << (e.strand == strnd::FWD ? "+" : "-")
              << sep << refSketch.metadata[e.refSeqId].name << sep << refSketch.metadata[e.refSeqId].len
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
};

}  // namespace skch

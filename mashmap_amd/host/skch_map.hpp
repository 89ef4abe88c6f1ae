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
#include "skch_map_post.hpp"
#include "skch_sketch.hpp"
#include "skch_types.hpp"

namespace skch {

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
  MapPost post;                                  // everything downstream of the device integers (skch_map_post.hpp)
  struct Batch {
    std::string bases;
    std::vector<int64_t> offs{0};
    std::vector<std::string> names;
    seqno_t firstSeqCounter = 0;
    void clear(seqno_t next) { bases.clear(); offs.assign(1, 0); names.clear(); firstSeqCounter = next; }
    size_t size() const { return names.size(); }
  };
  [[noreturn]] void die(const char* what) const {
    std::cerr << "[mashmap_hip::skch::Map] ERROR: " << what << ": " << mm_last_error(ctx) << std::endl;
    exit(1);
  }

 public:
  Map(const skch::Parameters& p, const skch::Sketch& refsketch, PostProcessResultsFn_t f = nullptr)
      : param(p), refSketch(refsketch), processMappingResults(f),
        refIdGroup(p.skip_prefix ? refsketch.refGroups() : std::vector<int>(refsketch.metadata.size(), 0)), ctx(refsketch.ctx()),
        post(p, refsketch.metadata, refIdGroup) {
    post.qmetadata = &qmetadata;
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
        post.filterByGroup(tmp, filtered, n_mappings, true);
        tmp.clear();
        b = e;
      }
      allReadMappings = std::move(filtered);
      std::sort(allReadMappings.begin(), allReadMappings.end(), [](const MappingResult& a, const MappingResult& b2) {
        return std::tie(a.querySeqId, a.queryStartPos, a.refSeqId, a.refStartPos) < std::tie(b2.querySeqId, b2.queryStartPos, b2.refSeqId, b2.refStartPos);
      });
      std::ostringstream os;
      post.reportReadMappings(allReadMappings, "", os);
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
        post.mapModule(D, r, batch.names[r], len, batch.firstSeqCounter + (seqno_t)r, perRead[r]);
        if (reportNow && !perRead[r].empty()) { os.str(std::string()); post.reportReadMappings(perRead[r], batch.names[r], os); text[r] = os.str(); }
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

};

}  // namespace skch

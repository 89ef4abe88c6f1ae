// mashmap_amd/host/skch_map.hpp -- skch::Map on top of the C ABI (include/mashmap_hip.h).
//
// Stands in for the reference's class of the same name (src/map/include/computeMap.hpp:53-1820): same constructor
// signature (maps every query and writes param.outFileName in the ctor, :123-141), same optional per-mapping callback
// (:100-101, called once per reported MappingResult, :1802), same PAF text (:1758-1806).
//
// What moved: sketchSequence, getSeedHits, getSeedIntervalPoints, computeL1CandidateRegions and
// computeL2MappedRegions (:818-1451) and the best-first / early-exit walk of doL2Mapping (:1182-1267, k_l2_select on integer tables
// of its two float decisions) run on the GPU for a whole batch of reads at once (mm_map_fragments).  What stays here, on host
// threads, is everything that touches floating point or std:: algorithm order:
//   * the floats of MappingResult (nucIdentity, its upper bound, kmerComplexity) from the candidate mappings' integers,
//   * chaining (mergeMappingsInRange :1580-1702), filterWeakMappings :423, the plane-sweep filters
//     (filter.hpp:103-160 query axis, :334-396 reference axis), filterFalseHighIdentity :441,
//     mappingBoundarySanityCheck :1714, sparsifyMappings :481 and the PAF writer.
// The reference runs one pthread task per read (ThreadPool.hpp); here three stages run concurrently on successive batches
// (MASHMAP_HIP_BATCH_MBP, default 512 Mbp): the reader thread parses batch i+2, the device stage maps batch i+1, the post stage
// chains / filters / prints batch i on param.threads std::threads.  Output order == input order (ThreadPool.hpp:187-211).
// A device PASS covers as many parsed batches as it takes to fill the GPU (MASHMAP_HIP_COALESCE_MBP, default 3072 Mbp; the kernels of a
// 512 Mbp pass run at ~120 Gbp/s, those of a 2 Gbp pass at ~150): the batches of a pass are laid end to end in HBM by
// mm_reads_upload_packed_parts, each from its own page-locked buffer, so the reader's unit (and the memory it locks) stays small.  A
// pass takes what is queued when the GPU falls free and never waits for more (greedy; MASHMAP_HIP_PASS_POLICY=ramp: the round-4 plan that
// grew passes 1, 1, 2, 4 ... to the coalescing limit and shrank them towards a known end).
//
// Device stage.  The kernels report, per fragment, the candidate mappings doL2Mapping would have pushed (mm_mapping, k_l2_select);
// with several contexts (MASHMAP_HIP_DEVICES, one per GPU, index replicated by Sketch) a batch -- 512 Mbp PER CONTEXT -- is cut into
// contiguous blocks of about equal bases, every context maps its block and hands its records to the host (each GPU over its own
// link; rank-major == input order) before the CPU filters, the one-to-one filter (:358-405) included, see them.
// MASHMAP_HIP_EXCHANGE=allgather puts the device-side all-gatherv (mm_allgatherv_mappings_local, RCCL over xGMI) in front instead.
#pragma once
#include <malloc.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <memory>
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
#include "pass_plan.hpp"
#include "seq_parse.hpp"
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
  std::vector<mm_ctx*> ctxs;                     // one per GPU (Sketch::contexts)
  mm_ctx* ctx;                                   // ctxs[0]
  MapPost post;                                  // everything downstream of the device integers (skch_map_post.hpp)
  std::unique_ptr<mmhost::WorkerPool> postPool;  // the post stage's threads
  bool packedUpload = true;                      // batches travel as 2-bit codes + N mask (set in mapQuery)
  // Host-to-device copies run ahead of the kernels: every context has a staging area that takes the blocks of parsed batches in queue
  // order (mm_reads_prefetch_packed_append), up to twice a pass's worth.  Whoever sees room sends the next block -- the reader the moment
  // it has parsed a batch, the device stage right after an upload has emptied part of the area -- so the copies of the batches behind a
  // pass always run under that pass's kernels.  stagedBases[i]: bases of queued batches already sent to context i (guarded by pfMu).
  std::mutex pfMu; std::vector<size_t> stagedBases; size_t stageCapBases = 0, stageReserveBases = 0; bool earlyPrefetch = true;
  bool exchangeFellBack = false;                 // a default RCCL all-gatherv failed once: per-context downloads for the rest of the run
  skch::Time::time_point tStart = skch::Time::now();   // MASHMAP_HIP_TIMING lines carry the time since the Map was constructed
  // one write per diagnostic line: three stages log at once, and `std::cerr << a << b` from two threads interleaves inside a line
  struct LogLine {
    std::ostringstream os;
    template <class T> LogLine& operator<<(const T& v) { os << v; return *this; }
    ~LogLine() { os << '\n'; const std::string t = os.str(); std::fwrite(t.data(), 1, t.size(), stderr); }
  };
  std::string at() const { char b[48]; snprintf(b, sizeof b, " [t=%.4f]", std::chrono::duration<double>(skch::Time::now() - tStart).count()); return b; }
  struct Batch {
    mmhost::ParsedBatch in;                      // names, offsets, bases (page-locked buffer, recycled through bufferPool)
    seqno_t firstSeqCounter = 0;
    // candidate mappings of the batch, read-major: a slice of the records its device pass brought back (page-locked, skch_types.hpp;
    // shared by the batches of the pass)
    std::shared_ptr<PinnedRecs<mm_mapping>> recOwner; const mm_mapping* recs = nullptr; size_t nRecs = 0;
    mutable std::vector<char> prefetched;        // per context: the batch's block is already on its way to that GPU (guarded by pfMu)
    size_t size() const { return in.names.size(); }
    size_t bases() const { return (size_t)in.totalBases(); }
  };
  // page-locked batch buffers come from skch::HostBufferPool (allocated while the index was being built) and are recycled
  [[noreturn]] void die(const char* what, mm_ctx* c = nullptr) const {
    std::cerr << "[mashmap_hip::skch::Map] ERROR: " << what << ": " << mm_last_error(c ? c : ctx) << std::endl;
    exit(1);
  }
  typedef mmhost::BatchChannel<Batch> Channel;     // hand-over between two stages (pass_plan.hpp)

 public:
  Map(const skch::Parameters& p, const skch::Sketch& refsketch, PostProcessResultsFn_t f = nullptr)
      : param(p), refSketch(refsketch), processMappingResults(f),
        refIdGroup(p.skip_prefix ? refsketch.refGroups() : std::vector<int>(refsketch.metadata.size(), 0)), ctxs(refsketch.contexts()),
        ctx(refsketch.ctx()), post(p, refsketch.metadata, refIdGroup) {
    post.qmetadata = &qmetadata;
    for (size_t i = 0; i < refsketch.metadata.size(); i++) refNameToId.emplace(refsketch.metadata[i].name, (int)i);
    // integer tables the kernels consume: estimateMinimumHitsRelaxed per Q.sketchSize (:1144), sketchCutoffs (:178-258), and the two
    // tables of doL2Mapping's walk (acceptance :1221, ANI cut-off :1192-1202)
    std::vector<int32_t> minHits = mmhost::minHitsTable(p.sketchSize, p.kmerSize, p.percentageIdentity);
    std::vector<int> cut = mmhost::sketchCutoffs(p.sketchSize, p.kmerSize, p.ANIDiff, p.ANIDiffConf, p.stage1_topANI_filter);
    std::vector<int32_t> cut32(cut.begin(), cut.end());
    std::vector<uint8_t> accept; std::vector<int16_t> minIsz;
    mmhost::replayTables(p.sketchSize, p.kmerSize, p.percentageIdentity, p.ANIDiff, p.keep_low_pct_id,
                         mmhost::availableCpus(), accept, minIsz);
    for (mm_ctx* c : ctxs) {
      if (mm_set_tables(c, minHits.data(), minHits.size(), cut32.data(), cut32.size()) != MM_OK) die("mm_set_tables", c);
      if (mm_set_replay_tables(c, accept.data(), minIsz.data(), (size_t)p.sketchSize + 1) != MM_OK) die("mm_set_replay_tables", c);
    }
    if (getenv("MASHMAP_HIP_TIMING")) LogLine() << "[mashmap_hip::timing] integer tables ready" << at();
    this->mapQuery();
    if (getenv("MASHMAP_HIP_TIMING")) LogLine() << "[mashmap_hip::timing] mapQuery done" << at();
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
    // a batch = MASHMAP_HIP_BATCH_MBP (default 512 Mbp) PER CONTEXT: every GPU of a sharded run gets a block as large as the batch of a
    // single-GPU run (tens of milliseconds of kernels), instead of an n-th of it
    const QueryBatchPlan plan = queryBatchPlan(param.querySequences, ctxs.size());
    const size_t batchBases = plan.batchBases;
    packedUpload = getenv("MASHMAP_HIP_ASCII_UPLOAD") == nullptr;
    // batches per device pass: only with one context (the blocks of a sharded batch are not consecutive reads across batches) and packed
    // uploads; MASHMAP_HIP_COALESCE_MBP=0 maps every batch by itself
    const size_t maxGroup = (ctxs.size() == 1 && packedUpload) ? std::max<size_t>(1, plan.passBases / std::max<size_t>(1, batchBases)) : 1;
    Channel parsed(std::max<size_t>(2, maxGroup)), mapped(std::max<size_t>(2, 2 * maxGroup));
    if (maxGroup > 1) {
      // passes grow from one batch to maxGroup: the first pass sizes the contexts' staging buffers for the largest one (a quarter of head
      // room for the overlapping tail fragments of reads that are not a multiple of segLength long), the others then launch against them
      const uint64_t frags = plan.passBases / (uint64_t)std::max<offset_t>(1, param.segLength);
      const int reserve = (int)std::min<uint64_t>(0x7fffffffu, frags + frags / 4 + 1024);
      for (mm_ctx* c : ctxs) if (mm_set_option(c, MM_OPT_RESERVE_FRAGMENTS, reserve) != MM_OK) die("mm_set_option", c);
    }
    if (!getenv("MASHMAP_HIP_NO_MALLOPT") && (!plan.inputKnown || plan.inputBytes > (256u << 20))) {
      // every batch allocates and frees a few megabyte-sized vectors (records, per-read results, PAF text) from three stages at once:
      // keep them on the heap instead of mmap/munmap per batch (each unmap interrupts every thread of the process), and keep the heap
      mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 256 << 20);
    }
    // the reader's workers normalise and pack the bases (2 bit + N mask, pack2bit.hpp) while they drop the line breaks, so that PCIe
    // carries 0.375 bytes per base instead of 1 (mm_reads_upload_packed); MASHMAP_HIP_ASCII_UPLOAD=1 ships ASCII to k_pack2bit instead
    earlyPrefetch = getenv("MASHMAP_HIP_NO_EARLY_PREFETCH") == nullptr;
    stagedBases.assign(ctxs.size(), 0);
    stageCapBases = stagingCapBases(plan, ctxs.size());                               // per context: the pass being assembled and the one behind it (skch_types.hpp)
    stageReserveBases = stagingReserveBases(plan, ctxs.size());
    // diagnostic (MASHMAP_HIP_STALL_TRACE=1): a thread that sleeps 0.5 ms at a time and reports when the sleep, a one-page mmap/munmap
    // (address-space lock) or a first touch of a fresh page took more than 3 ms -- tells a process-wide stall (scheduler, CPU quota)
    // from a lock inside the process when the stage timings show all three stages pausing at once
    std::atomic<bool> stallStop{false};
    std::thread stallTrace;
    if (getenv("MASHMAP_HIP_STALL_TRACE")) stallTrace = std::thread([&]() {
      while (!stallStop) {
        const auto a = skch::Time::now();
        std::this_thread::sleep_for(std::chrono::microseconds(500));
        const auto b = skch::Time::now();
        void* q = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        const auto c = skch::Time::now();
        if (q != MAP_FAILED) { *(volatile char*)q = 1; }
        const auto d = skch::Time::now();
        if (q != MAP_FAILED) munmap(q, 4096);
        const auto e = skch::Time::now();
        auto sec = [](skch::Time::time_point x, skch::Time::time_point y) { return std::chrono::duration<double>(y - x).count(); };
        if (sec(a, e) > 0.0035)
          LogLine() << "[mashmap_hip::stall] sleep " << sec(a, b) << " mmap " << sec(b, c) << " touch " << sec(c, d) << " munmap " << sec(d, e) << at();
      }
    });
    std::thread reader([&]() {
      // multi-threaded ingest (seq_parse.hpp): a window of the file per batch, parsed straight into a page-locked buffer.  8 workers:
      // memchr + memcpy at that width keep up with the device stage, and more of them page-faulting through the same file mapping next
      // to the post stage's threads stall each other for tens of milliseconds at a time (profiles/r03d_e2e_thread_sweep.txt: 32 reader
      // threads 18 Gbp/s end to end, 8 threads 24-26)
      const char* rte = getenv("MASHMAP_HIP_READER_THREADS");
      const unsigned readerThreads = rte ? (unsigned)std::max(1, atoi(rte)) : std::min((unsigned)std::min(12, std::max(1, param.threads)), std::max(1u, mmhost::availableCpus() * 3 / 4));
      mmhost::BatchReader rd(param.querySequences, batchBases, readerThreads, {}, "",
                             [](size_t n) {                       // only when the pool (skch_sketch.hpp) has run dry
                               const auto t0 = skch::Time::now();
                               char* p = (char*)mm_host_alloc(n);
                               if (getenv("MASHMAP_HIP_TIMING")) LogLine() << "[mashmap_hip::timing] reader: page-locked " << n << " bytes itself in "
                                                                           << std::chrono::duration<double>(skch::Time::now() - t0).count() << " s";
                               return p;
                             }, [](char* p) { mm_host_free(p); }, packedUpload);
      while (true) {
        Batch batch;
        { auto b = HostBufferPool::instance().take(0); batch.in.bases = b.first; batch.in.cap = b.second; }
        const auto tr0 = skch::Time::now();
        const bool more = rd.next(batch.in);
        if (more && getenv("MASHMAP_HIP_TIMING")) LogLine() << "[mashmap_hip::timing] reader: parsed " << batch.size() << " records, " << batch.in.totalBases() << " bases in "
                                                            << std::chrono::duration<double>(skch::Time::now() - tr0).count() << " s" << at();
        if (!more) { HostBufferPool::instance().give(batch.in.bases, batch.in.cap); break; }
        batch.firstSeqCounter = seqCounter;
        for (size_t r = 0; r < batch.size(); r++) {
          if (batch.in.offs[r + 1] - batch.in.offs[r] > (int64_t)std::numeric_limits<offset_t>::max()) {       // see Sketch::build: no LARGE_CONTIG variant
            std::cerr << "[mashmap::skch::Map::mapQuery] ERROR: query sequence " << batch.in.names[r] << " has " << (batch.in.offs[r + 1] - batch.in.offs[r])
                      << " bp; sequences of more than " << std::numeric_limits<offset_t>::max()
                      << " bp need the reference's LARGE_CONTIG build (64-bit offset_t), which mashmap_hip does not provide" << std::endl;
            exit(1);
          }
          const offset_t len = (offset_t)(batch.in.offs[r + 1] - batch.in.offs[r]);
          totalBp += (uint64_t)len;
          if (param.filterMode == filter::ONETOONE) qmetadata.push_back(ContigInfo{batch.in.names[r], len});
          if (len < param.kmerSize) {
            std::cerr << std::endl << "WARNING, skch::Map::mapQuery, read " << batch.in.names[r] << " of " << len << "bp "
                      << " is not long enough for mapping at segment length " << param.segLength << std::endl;
          } else {
            totalReadsPickedForMapping++;
          }
          // short reads travel too (they yield no fragment) so that seqCounter == firstSeqCounter + index inside the batch
          seqCounter++;
        }
        if (batch.size()) {
          // room in the staging areas: this batch's blocks start travelling now.  Prefetch and hand-over are ONE step under pfMu -- the
          // device stage tops the areas up from the queue under the same lock, in queue order, so a block is never sent twice or out of
          // order.  The wait for room in the queue comes first: the device stage takes pfMu between two gets, so a put() that blocked
          // while holding it would never be served.
          parsed.waitSpace();
          std::lock_guard<std::mutex> lk(pfMu);
          if (earlyPrefetch && packedUpload) {
            const std::vector<size_t> cut = blocksOf(batch);
            for (size_t i = 0; i < ctxs.size(); i++) issuePrefetch(batch, i, cut);
          }
          parsed.put(std::move(batch));
        }
      }
      parsed.close();
    });
    std::thread poster([&]() {
      Batch cur;
      while (mapped.get(cur)) {
        postStage(cur, allReadMappings, totalReadsMapped, outstrm);
        if (cur.in.bases) { HostBufferPool::instance().give(cur.in.bases, cur.in.cap); cur.in.bases = nullptr; }   // (normally handed back by the device stage, right behind the upload)
        cur = Batch();
      }
    });
    {
      // pass size: ramps up from one batch (so the post stage has work after one batch's worth of time), levels at the coalescing limit,
      // and comes down again towards the end of an input whose size is known (the last pass is followed by nothing that could hide it)
      // Default: greedy -- a pass takes whatever the reader has queued when the GPU falls free (at least one batch, at most maxGroup) and
      // never waits for more.  A one-batch pass costs more per base than the reader needs for a batch, so the queue grows and the next pass
      // takes two or three: the sizes balance themselves where the device keeps up with the reader, the post stage gets its work in a
      // steady trickle instead of six batches at a time, and what is left when the input ends is one small pass.
      // MASHMAP_HIP_PASS_POLICY=ramp is the previous plan (1, 1, 2, 4 ... up to the coalescing limit, down again towards a known end).
      const char* ppe = getenv("MASHMAP_HIP_PASS_POLICY");
      const bool greedy = !(ppe && std::string(ppe) == "ramp");
      std::vector<Batch> grp;
      uint64_t doneBases = 0;
      while (true) {
        const size_t want = maxGroup == 1 ? 0 : greedy ? 1 : mmhost::passWant(batchBases, plan.passBases, plan.inputKnown, plan.inputBytes, doneBases);
        if (!parsed.getGroup(grp, want, maxGroup, greedy && maxGroup > 1)) break;
        deviceStage(grp, parsed);
        for (auto& b : grp) { doneBases += b.bases(); mapped.put(std::move(b)); }
        grp.clear();
      }
      mapped.close();
    }
    reader.join();
    poster.join();
    stallStop = true; if (stallTrace.joinable()) stallTrace.join();
    HostBufferPool::instance().stop();                      // nobody asks for page-locked buffers any more

    if (param.filterMode == filter::ONETOONE) {            // :358-406
      const auto tf0 = skch::Time::now();
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
      std::string text;
      post.appendReadMappings(allReadMappings, "", text);
      outstrm.write(text.data(), (std::streamsize)text.size());
      if (processMappingResults) for (const auto& e : allReadMappings) processMappingResults(e);
      if (getenv("MASHMAP_HIP_TIMING")) LogLine() << "[mashmap_hip::timing] one-to-one filter + output " << std::chrono::duration<double>(skch::Time::now() - tf0).count() << " s" << at();
    }
    std::cerr << "[mashmap::skch::Map::mapQuery] count of mapped reads = " << totalReadsMapped
              << ", reads qualified for mapping = " << totalReadsPickedForMapping << ", total input reads = " << seqCounter
              << ", total input bp = " << totalBp << std::endl;
  }

  // sends context i's block of `b` ahead into that context's staging area, if there is room (pfMu held)
  void issuePrefetch(const Batch& b, size_t i, const std::vector<size_t>& cut) {
    if (b.prefetched.size() != ctxs.size()) b.prefetched.assign(ctxs.size(), 0);
    if (b.prefetched[i] || !b.in.packed) return;
    const int64_t o0 = b.in.packOffs[cut[i]], o1 = b.in.packEnd(cut[i], cut[i + 1]);
    const size_t blockBases = (size_t)(b.in.offs[cut[i + 1]] - b.in.offs[cut[i]]);
    if (o1 <= o0 || stagedBases[i] + blockBases > stageCapBases) return;
    mm_ctx* c = ctxs[i];
    const size_t reserve = stageReserveBases;                                          // what skch::Sketch has allocated behind the index build
    int staged = 0;
    if (mm_reads_prefetch_packed_append(c, b.in.bases2() + o0 / 16, b.in.nmask() + o0 / 32, (size_t)(o1 - o0), reserve, &staged) != MM_OK) die("mm_reads_prefetch_packed_append", c);
    if (!staged) return;                                    // the ring had no room under the pieces on their way: asked again when an upload has emptied part of it
    b.prefetched[i] = 1; stagedBases[i] += blockBases;
  }

  // ------------------------------------------------------------------------------------------------------------------
  // device stage: the reads of a pass (one batch; several with a single context) -> candidate mappings, on one GPU or sharded over all
  // contexts in contiguous blocks of about equal bases, one per context (a block may be empty)
  std::vector<size_t> blocksOf(const Batch& batch) const {
    const size_t nReads = batch.size(), nCtx = ctxs.size();
    std::vector<size_t> cutAt(nCtx + 1, nReads);
    cutAt[0] = 0;
    const int64_t total = batch.in.offs[nReads];
    size_t r = 0;
    for (size_t i = 1; i < nCtx; i++) {
      const int64_t want = total * (int64_t)i / (int64_t)nCtx;
      while (r < nReads && batch.in.offs[r] < want) r++;
      cutAt[i] = r;
    }
    return cutAt;
  }

  void deviceStage(std::vector<Batch>& grp, Channel& parsed) {
    const size_t nCtx = ctxs.size(), nB = grp.size();
    const bool timing = getenv("MASHMAP_HIP_TIMING") != nullptr;
    const auto t0 = skch::Time::now();
    size_t nReads = 0, passBases = 0;
    for (const auto& b : grp) { nReads += b.size(); passBases += b.bases(); }
    // per read of the pass (batches end to end): the reference group of its name, the reference contig of the same name
    std::vector<int32_t> readGroup, readSelf;
    std::vector<size_t> first(nB + 1, 0);
    for (size_t j = 0; j < nB; j++) first[j + 1] = first[j] + grp[j].size();
    if (param.skip_prefix) { readGroup.resize(nReads); for (size_t j = 0; j < nB; j++) for (size_t r = 0; r < grp[j].size(); r++) readGroup[first[j] + r] = getRefGroup(grp[j].in.names[r]); }
    if (param.skip_self) {
      readSelf.resize(nReads);
      for (size_t j = 0; j < nB; j++) for (size_t r = 0; r < grp[j].size(); r++) { auto it = refNameToId.find(grp[j].in.names[r]); readSelf[first[j] + r] = it == refNameToId.end() ? -1 : it->second; }
    }
    // several batches per pass only with one context (mapQuery): the blocks are then whole batches and the pass's reads are consecutive
    std::vector<std::vector<size_t>> cutAt(nB);
    for (size_t j = 0; j < nB; j++) cutAt[j] = blocksOf(grp[j]);
    // How the blocks' candidate mappings reach the host stage.  With a GPU per context (MASHMAP_HIP_DEVICES names distinct devices) the
    // exchange step is the one north_star describes: the RCCL all-gatherv of the candidate mappings over xGMI
    // (mm_allgatherv_mappings_local: count all-gather + one grouped broadcast per rank) in front of a single download from context 0 --
    // the layout one-process-per-GPU runs use (bench.py --gpus N, mm_allgatherv_mappings_begin/_end).  Contexts that share a GPU cannot
    // form an RCCL communicator: there every context downloads its own block (rank-major concatenation == input order), which serves
    // every filter mode just as well -- the one-to-one filter (:358-405) runs on the host over all records either way.
    // MASHMAP_HIP_EXCHANGE=allgather | download picks one explicitly; a default all-gatherv that RCCL refuses falls back to the
    // downloads with one warning (an explicit one is fatal).
    const char* xe = getenv("MASHMAP_HIP_EXCHANGE");
    const bool gatherAsked = xe && std::string(xe) == "allgather";
    if (gatherAsked && nCtx > 1 && !refSketch.commReady()) die("MASHMAP_HIP_EXCHANGE=allgather, but the contexts have no communicator (mm_comm_init_local failed)");
    bool gatherOnDevice = nCtx > 1 && !exchangeFellBack && refSketch.commReady() && (xe ? gatherAsked : refSketch.distinctDevices());
    std::vector<PinnedRecs<mm_mapping>> blockRecs(nCtx == 1 ? 0 : nCtx);
    auto all = std::make_shared<PinnedRecs<mm_mapping>>();
    double phase[3] = {0, 0, 0};                           // context 0: upload, kernels, download (seconds)
    auto releaseBuffers = [&]() {                           // the bases are in HBM: the page-locked buffers go back to the reader
      for (auto& b : grp) if (b.in.bases) { HostBufferPool::instance().give(b.in.bases, b.in.cap); b.in.bases = nullptr; }
    };
    auto runBlock = [&](size_t i) {
      mm_ctx* c = ctxs[i];
      const auto p0 = skch::Time::now();
      const seqno_t seqBase = grp[0].firstSeqCounter + (seqno_t)cutAt[0][i];
      if (grp[0].in.packed) {
        std::vector<mm_packed_part> parts(nB);
        for (size_t j = 0; j < nB; j++) {
          const Batch& bt = grp[j];
          const size_t b = cutAt[j][i], e = cutAt[j][i + 1];
          const int64_t o = bt.in.packOffs[b];
          parts[j] = mm_packed_part{bt.in.bases2() + o / 16, bt.in.nmask() + o / 32, bt.in.hasN.data() + b, bt.in.lens.data() + b, bt.in.packOffs.data() + b, e - b,
                                    param.skip_prefix ? readGroup.data() + first[j] + b : nullptr, param.skip_self ? readSelf.data() + first[j] + b : nullptr};
        }
        if (mm_reads_upload_packed_parts(c, parts.data(), nB, seqBase) != MM_OK) die("mm_reads_upload_packed_parts", c);
      } else {                                              // MASHMAP_HIP_ASCII_UPLOAD: one batch per pass, packed on the device
        const size_t b = cutAt[0][i], e = cutAt[0][i + 1];
        if (mm_reads_upload(c, grp[0].in.bases, grp[0].in.offs.data() + b, e - b, param.skip_prefix ? readGroup.data() + b : nullptr,
                            param.skip_self ? readSelf.data() + b : nullptr, seqBase) != MM_OK) die("mm_reads_upload", c);
      }
      if (nCtx == 1) releaseBuffers();
      {
        // this context's staging area has room again: the batches the reader has parsed behind this pass start travelling, in queue
        // order, while the pass is mapped; what the reader parses from now on it sends itself
        std::lock_guard<std::mutex> lk(pfMu);
        for (const auto& bt : grp) if (bt.prefetched.size() == nCtx && bt.prefetched[i]) { const auto ct = blocksOf(bt); stagedBases[i] -= (size_t)(bt.in.offs[ct[i + 1]] - bt.in.offs[ct[i]]); }
        if (earlyPrefetch && packedUpload) parsed.forEach([&](Batch& q) { issuePrefetch(q, i, blocksOf(q)); });
      }
      const auto p1 = skch::Time::now();
      if (mm_map_fragments(c) != MM_OK) die("mm_map_fragments", c);
      if (i == 0) { phase[0] = std::chrono::duration<double>(p1 - p0).count(); phase[1] = std::chrono::duration<double>(skch::Time::now() - p1).count(); }
      if (!blockRecs.empty() && !gatherOnDevice) {
        size_t nb = 0;
        if (mm_mappings_count(c, &nb) != MM_OK) die("mm_mappings_count", c);
        blockRecs[i].resize(nb);
        if (nb && mm_mappings_download(c, blockRecs[i].data(), nb, &nb) != MM_OK) die("mm_mappings_download", c);
      }
    };
    if (nCtx == 1) runBlock(0);
    else {
      std::vector<std::thread> th;
      for (size_t i = 1; i < nCtx; i++) th.emplace_back(runBlock, i);
      runBlock(0);
      for (auto& t : th) t.join();
      releaseBuffers();
    }
    size_t n = 0;
    const auto p2 = skch::Time::now();
    if (nCtx == 1) {
      if (mm_mappings_count(ctx, &n) != MM_OK) die("mm_mappings_count");
      all->resize(n);
      if (n && mm_mappings_download(ctx, all->data(), n, &n) != MM_OK) die("mm_mappings_download");
    }
    if (nCtx > 1 && gatherOnDevice) {
      if (mm_allgatherv_mappings_local(ctxs.data(), (int)nCtx) != MM_OK) {
        if (gatherAsked) die("mm_allgatherv_mappings_local");
        std::cerr << "[mashmap_hip::skch::Map] WARNING: the RCCL all-gatherv of the candidate mappings failed (" << mm_last_error(ctx)
                  << "); every context downloads its own block from here on" << std::endl;
        exchangeFellBack = true; gatherOnDevice = false;
        for (size_t i = 0; i < nCtx; i++) {
          size_t nb = 0;
          if (mm_mappings_count(ctxs[i], &nb) != MM_OK) die("mm_mappings_count", ctxs[i]);
          blockRecs[i].resize(nb);
          if (nb && mm_mappings_download(ctxs[i], blockRecs[i].data(), nb, &nb) != MM_OK) die("mm_mappings_download", ctxs[i]);
        }
      } else {
        if (mm_gathered_counts(ctx, nullptr, &n) != MM_OK) die("mm_gathered_counts");
        all->resize(n);
        if (n && mm_gathered_download(ctx, all->data(), n) != MM_OK) die("mm_gathered_download");
      }
    }
    if (nCtx > 1 && !gatherOnDevice) {
      for (const auto& v : blockRecs) n += v.size();
      all->resize(n);
      size_t at = 0;
      for (const auto& v : blockRecs) { if (!v.empty()) std::memcpy(all->data() + at, v.data(), v.size() * sizeof(mm_mapping)); at += v.size(); }
    }
    // the records are read-major in input order: every batch of the pass gets its slice
    {
      size_t at = 0;
      for (size_t j = 0; j < nB; j++) {
        const seqno_t endId = grp[j].firstSeqCounter + (seqno_t)grp[j].size();
        const size_t b0 = at;
        while (at < n && (*all)[at].querySeqId < endId) at++;
        grp[j].recOwner = all; grp[j].recs = all->data() + b0; grp[j].nRecs = at - b0;
      }
      if (at != n) { std::cerr << "[mashmap_hip::skch::Map] ERROR: candidate mappings of a pass are not in input order" << std::endl; exit(1); }
    }
    if (timing) LogLine() << "[mashmap_hip::timing] device stage (upload + pack + kernels" << (gatherOnDevice ? " + all-gatherv" : "") << " + download of " << n
                          << " candidate mappings): " << std::chrono::duration<double>(skch::Time::now() - t0).count() << " s (upload " << phase[0] << ", kernels " << phase[1]
                          << ", download " << std::chrono::duration<double>(skch::Time::now() - p2).count() << ") [bases " << passBases << "] [batches " << nB << "]" << at();
  }

  // post stage: per read, chaining + filters + PAF text on param.threads threads; then output in input order
  void postStage(Batch& batch, MappingResultsVector_t& allReadMappings, seqno_t& totalReadsMapped, std::ofstream& outstrm) {
    const size_t nReads = batch.size();
    const bool timing = getenv("MASHMAP_HIP_TIMING") != nullptr;
    const auto t0 = skch::Time::now();
    // the records are read-major: first record of every read
    std::vector<size_t> recBegin(nReads + 1, batch.nRecs);
    {
      size_t i = 0;
      for (size_t r = 0; r <= nReads; r++) {
        while (i < batch.nRecs && (size_t)(batch.recs[i].querySeqId - batch.firstSeqCounter) < r) i++;
        recBegin[r] = i;
      }
    }
    // per CHUNK of 64 consecutive reads, not per read: one text buffer, one result vector, one counter (a hundred thousand small
    // allocations and frees per batch on this thread cost as much as the stage's work)
    const bool reportNow = param.filterMode != filter::ONETOONE;
    const bool keepMaps = !reportNow || (bool)processMappingResults;
    const size_t chunk = 64, nChunks = (nReads + chunk - 1) / chunk;
    std::vector<std::string> chunkText(reportNow ? nChunks : 0);
    std::vector<MappingResultsVector_t> chunkMaps(keepMaps ? nChunks : 0);
    std::vector<int32_t> chunkMapped(nChunks, 0);
    // no wider than the CPUs the process may use (a container's quota: seq_parse.hpp availableCpus) -- more threads than that do not
    // finish the batch sooner, they get the whole process throttled
    const char* pte = getenv("MASHMAP_HIP_POST_THREADS");
    const unsigned nThreads = pte ? (unsigned)std::max(1, atoi(pte)) : std::min((unsigned)std::max(1, param.threads), mmhost::availableCpus());
    std::atomic<size_t> next(0);
    auto work = [&]() {
      std::string text;                                    // the chunk's PAF lines (MapPost::appendReadMappings: std::to_chars, no stream)
      MappingResultsVector_t one;
      for (size_t ci = next.fetch_add(1); ci < nChunks; ci = next.fetch_add(1)) {
        text.clear();
        int mappedHere = 0;
        for (size_t r = ci * chunk; r < std::min(nReads, (ci + 1) * chunk); r++) {
          const offset_t len = (offset_t)(batch.in.offs[r + 1] - batch.in.offs[r]);
          if (len < param.kmerSize || recBegin[r] == recBegin[r + 1]) continue;
          one.clear();
          post.mapModuleFromRecords(batch.recs + recBegin[r], batch.recs + recBegin[r + 1], len, one);
          if (one.empty()) continue;
          mappedHere++;
          if (reportNow) post.appendReadMappings(one, batch.in.names[r], text);
          if (keepMaps) chunkMaps[ci].insert(chunkMaps[ci].end(), one.begin(), one.end());
        }
        chunkMapped[ci] = mappedHere;
        if (reportNow) chunkText[ci] = text;
      }
    };
    if (!postPool || postPool->size() != nThreads) postPool.reset(new mmhost::WorkerPool(nThreads));   // persistent: a batch is milliseconds of work
    postPool->run(nThreads, [&](unsigned) { work(); });
    const auto t1 = skch::Time::now();
    size_t textBytes = 0;
    for (const auto& t : chunkText) textBytes += t.size();
    std::string all;                                       // the batch's PAF text, input order, written with one call
    all.reserve(textBytes);
    for (size_t ci = 0; ci < nChunks; ci++) {              // mapModuleHandleOutput (:724-752), input order
      totalReadsMapped += chunkMapped[ci];
      if (!reportNow) allReadMappings.insert(allReadMappings.end(), chunkMaps[ci].begin(), chunkMaps[ci].end());
      else {
        all += chunkText[ci];
        if (processMappingResults) for (const auto& e : chunkMaps[ci]) processMappingResults(e);
      }
    }
    if (!all.empty()) outstrm.write(all.data(), (std::streamsize)all.size());
    if (timing) LogLine() << "[mashmap_hip::timing] post stage: chain + filter + format " << std::chrono::duration<double>(t1 - t0).count()
                          << " s, output " << std::chrono::duration<double>(skch::Time::now() - t1).count() << " s" << at();
  }

};

}  // namespace skch

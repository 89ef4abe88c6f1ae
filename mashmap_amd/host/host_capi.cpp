// mashmap_amd/host/host_capi.cpp -- C entry points around skch::MapPost (skch_map_post.hpp), the host half of skch::Map that turns
// the device's candidate mappings into reported MappingResult rows (chaining computeMap.hpp:1580, filters filter.hpp:103,334,
// sanity checks :1714).  Built as mashmap_amd/lib/libmashmap_host.so so that bench.py can time the stage a user's run goes through
// after the kernels ("packed bases -> MappingResult rows") with the same code skch::Map runs, on all host cores.
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "skch_map_post.hpp"
#include "../csrc/mm_exchange_plan.h"

extern "C" {

// one reported mapping, integers + the two identities (what the PAF line is printed from)
struct mmh_row {
  int32_t querySeqId, queryLen, queryStartPos, queryEndPos, refSeqId, refStartPos, refEndPos, strand, conservedSketches, blockLength;
  float nucIdentity, kmerComplexity;
};

// per read: mapModule's tail (MapPost::mapModuleFromRecords) on `threads` threads.  recs: the batch's candidate mappings, read-major
// (mm_mappings_download / mm_gathered_download); readLens[r] for querySeqId == firstSeqCounter + r.  Returns the number of reported
// mappings (rows, if non-null, receives the first `cap` of them in input order); *seconds = wall time of the stage.
int64_t mmh_post_batch(int k, int segLength, int sketchSize, float pi, int filterMode, int hgFilter, int numMappings,
                       int nContigs, const int32_t* contigLens, const mm_mapping* recs, size_t nRecs, const int32_t* readLens,
                       size_t nReads, int32_t firstSeqCounter, int threads, double* seconds, mmh_row* rows, size_t cap) {
  skch::Parameters p;
  p.kmerSize = k; p.segLength = segLength; p.block_length = segLength; p.chain_gap = segLength; p.sketchSize = sketchSize;
  p.percentageIdentity = pi; p.filterMode = filterMode; p.stage1_topANI_filter = hgFilter != 0;
  p.numMappingsForSegment = (uint32_t)numMappings; p.numMappingsForShortSequence = (uint32_t)numMappings;
  std::vector<skch::ContigInfo> meta((size_t)nContigs);
  for (int i = 0; i < nContigs; i++) meta[i] = skch::ContigInfo{"c" + std::to_string(i), contigLens[i]};
  std::vector<int> grp((size_t)nContigs, 0);
  skch::MapPost post(p, meta, grp);
  const auto t0 = std::chrono::high_resolution_clock::now();
  std::vector<size_t> recBegin(nReads + 1, nRecs);
  { size_t i = 0; for (size_t r = 0; r <= nReads; r++) { while (i < nRecs && (size_t)(recs[i].querySeqId - firstSeqCounter) < r) i++; recBegin[r] = i; } }
  std::vector<skch::MappingResultsVector_t> perRead(nReads);
  std::atomic<size_t> next(0);
  auto work = [&]() {
    const size_t chunk = 64;
    for (size_t r0 = next.fetch_add(chunk); r0 < nReads; r0 = next.fetch_add(chunk))
      for (size_t r = r0; r < std::min(nReads, r0 + chunk); r++)
        if (readLens[r] >= k && recBegin[r] != recBegin[r + 1]) post.mapModuleFromRecords(recs + recBegin[r], recs + recBegin[r + 1], readLens[r], perRead[r]);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  int64_t n = 0;
  for (size_t r = 0; r < nReads; r++)
    for (const auto& e : perRead[r]) {
      if (rows && (size_t)n < cap)
        rows[n] = mmh_row{e.querySeqId, e.queryLen, e.queryStartPos, e.queryEndPos, e.refSeqId, e.refStartPos, e.refEndPos, (int32_t)e.strand,
                          e.conservedSketches, e.blockLength, e.nucIdentity, (float)e.kmerComplexity};
      n++;
    }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  return n;
}

// slots of the all-gatherv of candidate mappings (mm_exchange_plan.h, what mm_comm.hip places its RCCL broadcasts by): disp[world + 1]
uint64_t mmh_exchange_plan(const uint64_t* counts, int world, uint64_t* disp) { return mm_exchange_place(counts, world, disp); }

}  // extern "C"

// mashmap_amd/csrc/mm_internal.h -- shared between the translation units of libmashmap_hip.so.
// gfx950 (MI355X, wave64) only.  No CUDA, no portability macros.
#pragma once
#include <chrono>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include "../../include/mashmap_hip.h"

#define MM_WAVE 64

// ---------------------------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------------------------
inline double g_mmAllocSeconds = 0.0;       // wall time inside hipMalloc / hipFree of DevBuf::ensure (MASHMAP_HIP_TIMING reports it per sized pass; one stream-ordered batch per context)
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t ensure(size_t need) {          // grow-only; contents are NOT preserved
    if (need <= bytes) return hipSuccess;
    const auto t0 = std::chrono::steady_clock::now();
    // the new buffer first, with an eighth of head room; the old one is given up only once the new one is there -- or, when the device
    // cannot hold both, before a second attempt at the exact size (a failed growth leaves the buffer as it was whenever it can)
    void* q = nullptr;
    size_t cap = need + need / 8 + 256;
    hipError_t e = hipMalloc(&q, cap);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
      e = hipMalloc(&q, cap);
      if (e != hipSuccess) { (void)hipGetLastError(); cap = need + 256; e = hipMalloc(&q, cap); }
    } else if (p) {
      (void)hipFree(p);
    }
    if (e == hipSuccess) { p = q; bytes = cap; }
    g_mmAllocSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// fragment descriptor on the device
struct DFrag {
  int64_t base;      // absolute index of the fragment's first base in the packed buffer
  int32_t len;       // bases
  int32_t readId;
};

// device index (see DESIGN.md "data layout in HBM")

#define MM_OPEN_BLOCK_SHIFT 10

struct DeviceIndex {
  size_t nRec = 0, nKeys = 0, nPoints = 0, nContigs = 0, htCap = 0, nOpen = 0;
  // minmerIndex as the L2 event stream (see mm_build_device_index): per contig, one insert event per record at wpos and one
  // eviction event at wpos_end, merged by position
  DevBuf evKey;                // uint32 pos*2 + isInsert
  DevBuf evAux;                // uint32 insert: wpos_end | REV<<31 ; eviction: 0
  DevBuf evHash;               // uint64 hash of the record
  DevBuf contigOff;            // int64[nContigs+1] event offsets
  // records still open at every MM_OPEN_BLOCK-th position (wpos < B < wpos_end, index order), as insert events: the L2 pre-load of a
  // candidate starts from the list of its block instead of streaming a whole segLength of events (computeMap.hpp:1323-1338)
  DevBuf opKey, opAux, opHash; // as evKey / evAux / evHash
  DevBuf blockOff;             // int64[nBlocks+1] offsets into op*
  DevBuf evBlock;              // int64[nBlocks+1] first event at or behind the block start (block b of a contig: pos >= b << MM_OPEN_BLOCK_SHIFT)
  DevBuf contigBlock;          // int64[nContigs+1] first block of every contig
  DevBuf contigLen;            // int32[nContigs]
  DevBuf refGroup;             // int32[nContigs] (all 0 when unused)
  DevBuf htSlots;              // {uint64 key, uint64 val}[htCap]; val = offset<<24 | count<<1 | freq (one 16-byte slot per probe)
  DevBuf filter; uint64_t filterMask = 0;   // presence filter in front of htSlots: uint64 words, mm_filter_word / mm_filter_bits (mm_device.h); word mask, 0 = disabled
  // large tables (human-scale index): htSlots is placed in buckets of MM_TAG_BUCKET slots and fronted by one TAG BYTE per slot (0 = empty):
  // a probe reads the 16 tags of the seed's home bucket -- one 16-byte load out of an array 1/16 the size of the table -- and touches
  // the 16-byte slot only where a tag matches (~86 % of query seeds are absent from the index: sequencing errors)
  DevBuf htTags; bool tagged = false;
  DevBuf ptKeys;               // uint64[nPoints]: seqId<<33 | pos<<1 | (side==OPEN)
  DevBuf keys, keyOff, keyFreq; // the lookup map's key table in the order of ptKeys: uint64 key, uint64 first point (nKeys + 1), uint8 isFrequent
  bool ready = false;
};

struct mm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  mm_params P{};
  std::string err;

  // host mirrors needed for downloads in reference layout
  std::vector<mm_minmer> hMinmers;
  std::vector<mm_minmer> hMinmersAll;                   // minmerIndex before dropFreqSeedSet (only with MM_OPT_KEEP_FULL_INDEX: --saveIndex)
  bool keepFullIndex = false;
  std::vector<uint64_t> hKeys, hOffsets, hFreq;
  std::vector<mm_interval_point> hPoints;
  bool mirrorMinmers = false, mirrorMap = false;        // the host copies above are filled (a device-built index fills them on request)
  int32_t freqThreshold = 0x7fffffff;

  DeviceIndex idx;
  DevBuf dMinHits, dCutoffs; size_t nMinHits = 0, nCutoffs = 0;

  // resident reads
  size_t nReads = 0, nFrags = 0, nPackedBases = 0;
  int32_t seqCounterBase = 0, maxFragLen = 0;
  DevBuf dAscii, dReadSrcOff, dReadPackOff, dReadLen, dReadGroup, dReadSelf, dReadHasN;
  // mm_reads_prefetch: the next batch's ASCII bytes, copied on a stream of their own while the current batch is mapped
  DevBuf dAsciiNext; hipStream_t copyStream = nullptr; hipEvent_t copyDone = nullptr;
  const void* prefetchPtr = nullptr; const void* prefetchPtr2 = nullptr; size_t prefetchBytes = 0; bool prefetchValid = false, prefetchPacked = false;
  struct StagedPart { const void* b2; const void* nm; size_t nPacked; size_t off; };   // packed parts sent ahead (mm_reads_prefetch_packed[_append]): [codes | mask] at dAsciiNext + off
  std::vector<StagedPart> staged; size_t stagedBytes = 0;
  std::mutex prefetchMu;                                // mm_reads_prefetch* may come from another thread than the uploads (a reader thread): the staging state is shared
  DevBuf dBases2, dNmask, dFrags;
  std::vector<mm_fragment> hFrags;
  // batches parked beside the resident one (mm_reads_exchange): everything an upload leaves behind, swapped by pointer
  struct ParkedReads {
    size_t nReads = 0, nFrags = 0, nPackedBases = 0; int32_t seqCounterBase = 0, maxFragLen = 0; bool windowed = false;
    DevBuf dReadSrcOff, dReadPackOff, dReadLen, dReadGroup, dReadSelf, dReadHasN, dBases2, dNmask, dFrags;
    std::vector<mm_fragment> hFrags;
  };
  ParkedReads parked[MM_BATCH_SLOTS];

  // sketches: raw (a4) and after frequent-seed removal (a8)
  DevBuf dSkHash, dSkPos, dSkStrand, dSkCount;          // [nFrags*s] u64, int2, i8 ; [nFrags] u32
  DevBuf dHardList, dCounters, dSketchSpill;            // dSketchSpill: first / last / strand-sum arrays of k_sketch_hard<K, true> (large sketches)
  DevBuf dSketchTabs; int sketchTabsK = 0;        // strip-hasher tables for kmerSize sketchTabsK (built once, copied into LDS by every workgroup)
  DevBuf dQHash, dQStrand;                              // post-removal sketch (written only for fragments that lose a frequent seed)
  DevBuf dStats;                                        // mm_frag_stats[nFrags]
  DevBuf dPtOff, dPts; size_t ptsCap = 0;               // per-fragment offset (int64) + sorted keys
  DevBuf dPtKept;                                       // int32[nFrags]: points of a queued fragment that reach the L1 kernels (k_gather_points, k_filter_points)
  // --noSplit with reads longer than segLength (windowLen != 0, computeMap.hpp:933): the literal kernels' state
  bool windowed = false;                                // some resident fragment is longer than segLength
  DevBuf dPtIds, dWinFreq, dWinExt, dWinHeap, dWinKeys, dWinVals, dWinOffH, dWinOffT, dWinCntH, dWinCntT;
  DevBuf dL1, dL1b, dL1Cursors; size_t l1Cap = 0, nL1 = 0;   // dL1b: the region-filled buffer k_l1_compact reads from
  DevBuf dL1Off;                                        // int64[nFrags] first candidate of a fragment
  DevBuf dL1Regions;                                    // L1Regions: fill count + prefix position of the 64 output regions (k_l1_regions)
  DevBuf dL2; size_t l2Cap = 0, nL2 = 0;
  DevBuf dL2First, dL2Num;                              // per L1 candidate: first locus in dL2 (int64) and count (int32)
  // candidate mappings (mm_select.hip)
  DevBuf dAccept, dMinIsz; bool haveReplayTables = false;
  DevBuf dSelCnt, dSelOff, dSelHeap, dFragTab, dMappings; size_t nMappings = 0; bool fragTabStale = true;
  // multi-GPU exchange (mm_comm.hip)
  void* comm = nullptr; int commRank = 0, commWorld = 0; bool commCopy = false;   // commCopy: local group whose contexts share a device
  DevBuf dCommCounts, dGathered; std::vector<size_t> gatherCounts, gatherDisp; size_t nGathered = 0; bool gathered = false;
  // overlapped exchange (mm_allgatherv_mappings_begin / _end): a snapshot of the records, a stream of its own, the host thread that
  // runs the exchange while the caller maps the next batch
  DevBuf dGatherSrc; hipStream_t commStream = nullptr; std::thread gatherThread; int gatherRc = 0; std::string gatherErr;
  std::vector<DevBuf*> allBufs();
  DevBuf dL2Info, dL2Cnt, dL2Off, dL2Ops, dScanTmp, dL2Tmp, dL2Wide, dL2Exact, dL2Cells, dListB, dListC, dBigList, dMidList;     // L2 staging: per-candidate stream extents, op counts/offsets, located ops
  DevBuf dL2InitCells, dL2InitState;                                 // per candidate of a chunk: the SlideMapper state after the pre-load, as k_l2_locate leaves it for the sweeps
  DevBuf dL2Sort[4], dL2Order, dL2OrderPos;                          // candidates of a chunk in order of descending stream length (mm_order_desc)
  bool sketched = false, mapped = false;
  // steady state: the previous pass of this context went through and left every buffer sized (mm_launch_map); what it saw
  bool steadyOk = false, lastSteady = false; size_t prevBig = 0, candCap = 0, l2Chunks = 1, sizedFrags = 0; int prevLocap = 0, steadyFails = 0;   // sizedFrags: fragments of the last sized pass
  unsigned long long* hPass = nullptr;                  // page-locked: the counters of a pass as read back at its end
  size_t lastHard = 0;                                  // fragments the fast sketch kernel handed to the hard list in the last pass
  size_t prevMid = 0, lastMid = 0; bool midKnown = false;   // fragments k_lookup_mid took in the last sized pass (its grid in the steady-state passes behind it)
  size_t lastOps = 0, lastBig = 0;                      // L2 stream entries reserved / fragments queued for the HBM point path in the last pass
  size_t nSyncs = 0;                                    // host synchronisations inside the last mm_map_fragments (diagnostics: mm_pass_syncs)
  uint64_t nPasses = 0, nSteadyPasses = 0, nRedone = 0; // mm_map_fragments calls of this context: all, those that went through as steady-state passes, steady attempts redone the sized way
  bool keepFiltered = false;                            // MM_OPT_KEEP_POINTS = 2: ... and k_filter_points runs on them as it does on a queued fragment's (mm_points_download returns what it leaves)
  bool keepPoints = false;                              // mm_set_option(MM_OPT_KEEP_POINTS): route every fragment through the HBM point list
  size_t reserveFrags = 0;                              // mm_set_option(MM_OPT_RESERVE_FRAGMENTS): fragments of the largest batch the caller will upload; sized passes size for it

  // profiling
  bool profile = false;
  double kMs[MM_K_COUNT] = {0};
  uint64_t kLaunches[MM_K_COUNT] = {0};
  hipEvent_t evA = nullptr, evB = nullptr;              // mm_bench_hash_only's own pair
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evPool; size_t evUsed = 0;   // KernelTimer brackets recorded since the last mm_profile_collect
  std::vector<std::pair<int, size_t>> evPending;                              // (kernel, pool index)
};

#define MM_HIP(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                           \
      return MM_ERR_DEVICE;                                                                      \
    }                                                                                            \
  } while (0)

// RAII hipEvent bracket on the ctx stream.  The two events are only RECORDED here (a pair out of a pool that grows with the number of
// brackets in flight); their times are read by mm_profile_collect once the stream has been synchronised anyway -- measuring a pass does
// not add host waits to it (a steady-state pass stays at its one synchronisation with the per-kernel timers on).
struct KernelTimer {
  mm_ctx* c; int which; size_t idx = 0; bool on = false;
  KernelTimer(mm_ctx* c_, int w) : c(c_), which(w) {
    if (!c->profile) return;
    if (c->evUsed == c->evPool.size()) {
      // the pool is recycled by mm_profile_collect (every profiled entry point that synchronises calls it); a caller that records
      // thousands of brackets without one gets no more events than this, and a failing hipEventCreate switches the timers off
      if (c->evPool.size() >= 8192) return;
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { if (a) (void)hipEventDestroy(a); c->profile = false; return; }
      c->evPool.emplace_back(a, b);
    }
    idx = c->evUsed++;
    on = hipEventRecord(c->evPool[idx].first, c->stream) == hipSuccess;
  }
  ~KernelTimer() {
    if (!on) return;
    if (hipEventRecord(c->evPool[idx].second, c->stream) == hipSuccess) c->evPending.emplace_back(which, idx);
  }
};
// adds the recorded brackets to kMs / kLaunches; the stream must have been synchronised behind them
inline void mm_profile_collect(mm_ctx* c) {
  for (const auto& pr : c->evPending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->evPool[pr.second].first, c->evPool[pr.second].second) == hipSuccess) { c->kMs[pr.first] += ms; c->kLaunches[pr.first] += 1; }
  }
  c->evPending.clear(); c->evUsed = 0;
}

// MM_OPT_RESERVE_FRAGMENTS: a sized pass over a small batch sizes every staging buffer for the largest batch the caller has announced, so
// that the larger batches behind it are steady-state passes instead of being sized (and their buffers reallocated) again.
//   mm_frag_cap: fragments to size per-fragment buffers for;  mm_scaled: a count of this pass scaled to what the announced batch would bring
inline size_t mm_frag_cap(const mm_ctx* c, size_t nF) { return nF > c->reserveFrags ? nF : c->reserveFrags; }
// A scaled count never claims more than an eighth of the device memory that is free when it is asked for (`bytesPer` = bytes the buffer
// holds per counted item): the scaling is a convenience for the passes behind this one -- a repeat-rich first batch must not turn a pass
// that fits into an out-of-memory error; a later, larger batch that outgrows the clamped buffer is redone the sized way, as without scaling.
inline size_t mm_scaled(const mm_ctx* c, size_t count, size_t bytesPer) {
  const size_t nF = c->nFrags ? c->nFrags : 1;
  if (c->reserveFrags <= nF) return count;
  size_t scaled = (size_t)((double)count * (double)c->reserveFrags / (double)nF) + 1;
  size_t freeB = 0, totalB = 0;
  if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); return count; }
  const size_t most = freeB / 8 / (bytesPer ? bytesPer : 1);
  if (scaled > most) scaled = most > count ? most : count;
  return scaled;
}

// launchers implemented in the .hip files
int mm_check_params(const mm_params* p, std::string& err);
int mm_launch_pack(mm_ctx* c);
int mm_launch_pack_raw(mm_ctx* c, const uint8_t* dAscii, const int64_t* dSrcOff, const int64_t* dPackOff, const int32_t* dLen, int nReads,
                       int64_t nChunks, uint32_t* dB, uint32_t* dM, uint32_t* dHasN);
int mm_launch_sketch(mm_ctx* c);
int mm_launch_sketch_global(mm_ctx* c);   // mm_sketch_global.hip: sketches no LDS table holds (sketchSize > MM_LDS_MAX_SKETCH)
#define MM_LDS_MAX_SKETCH 8190             // beyond: the global-memory sketch kernel and the literal L2 kernels (any size up to MM_MAX_SKETCH)
#define MM_WINNOW_LDS_SKETCH 4096          // the device index build keeps a window's sketch as one sorted array in LDS up to here (k_winnow_tiles: an insert shifts half of it), and as
                                           // blocks of 64 entries in HBM under an LDS directory beyond (k_winnow_tiles<.., GSK>: 0.7 s against 8.7 s of index build at sketchSize 9 998,
                                           // profiles/r14_11; LDS itself would hold 10 000 entries)
#define MM_MAX_SKETCH 65535                // seeds are numbered in 16 bits where the literal kernels count windows per seed (mm_map.hip: ptIds); the reference takes any size
int mm_launch_map(mm_ctx* c);
// Steady state: once a pass of a context has sized every staging buffer, the next passes launch everything against
// those capacities with the counts left on the device and read ONE block of counters back at the end (one host synchronisation per
// pass); a pass that outgrows a buffer is detected there and redone the sized way (MM_PASS_REDO from the launchers).
#define MM_PASS_REDO 1
int mm_launch_select(mm_ctx* c, bool steady = false);
void mm_comm_release(mm_ctx* c);
int mm_launch_l2(mm_ctx* c, unsigned long long* cnt, bool steady = false);   // cnt: device counters [2] candidates [4] cursor [5] overflow [6] flags
int mm_build_device_index(mm_ctx* c, const int32_t* contigLen, const int32_t* refGroup, size_t nContigs);   // from the host mirrors
int mm_flatten_device_index(mm_ctx* c, const mm_minmer* dRec, size_t n, size_t nk, size_t np, const int32_t* contigLen, const int32_t* refGroup, size_t nContigs);
int mm_finalize_index_device(mm_ctx* c, const std::vector<std::pair<const mm_minmer*, size_t>>& parts, float kmerPctThreshold, const int32_t* contigLen,
                             const int32_t* refGroup, size_t nContigs);
int mm_mirror_minmers(mm_ctx* c);
int mm_mirror_map(mm_ctx* c);
int mm_scan_i32_to_i64(mm_ctx* c, int64_t n, const int32_t* dIn, int64_t* dOut, int64_t* total);
int mm_scan_i32_to_i64_dev(mm_ctx* c, int64_t n, const int32_t* dIn, int64_t* dOut, const int64_t** dTotal);   // the total stays on the device: no synchronisation
int mm_order_desc(mm_ctx* c, const int32_t* dKey, int c0, int n, int shift, int32_t* dOrder);   // mm_index_dev.hip (rocPRIM radix sort)
int mm_order_pairs(mm_ctx* c, int n, unsigned bits, int32_t* dOrder);                          // same file: pairs already in dL2Sort[0] / dL2Sort[2]

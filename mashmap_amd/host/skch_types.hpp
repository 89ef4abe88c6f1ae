// mashmap_amd/host/skch_types.hpp -- the data contracts the host side shares with its callers.
//
// Two build modes:
//   * inside the reference tree (-DMASHMAP_HIP_REFERENCE_TREE): the reference's own
//     src/map/include/base_types.hpp and map_parameters.hpp are used unchanged, so mash_map.cpp and
//     parseCmdArgs.hpp keep compiling against skch::Parameters / skch::MappingResult as they are;
//   * standalone (this repository, the GPU box): the same contracts are declared here with the same
//     names and field meaning (base_types.hpp:17-281, map_parameters.hpp:32-102), because the
//     reference sources do not travel.
#pragma once

#ifdef MASHMAP_HIP_REFERENCE_TREE
#include "map/include/base_types.hpp"
#include "map/include/map_parameters.hpp"
#else

#include <chrono>
#include <cstdint>
#include <filesystem>
#include <limits>
#include <string>
#include <vector>

namespace skch {

typedef uint64_t hash_t;
typedef int32_t offset_t;      // base_types.hpp:21 (the LARGE_CONTIG variant is not supported by the device layouts)
typedef int32_t seqno_t;
typedef int16_t strand_t;
typedef int8_t side_t;
typedef std::chrono::high_resolution_clock Time;

struct MinmerInfo {            // base_types.hpp:31 (24 bytes, same layout as mm_minmer)
  hash_t hash; offset_t wpos; offset_t wpos_end; seqno_t seqId; strand_t strand;
  bool operator<(const MinmerInfo& x) const { return seqId != x.seqId ? seqId < x.seqId : wpos < x.wpos; }
};
struct IntervalPoint {         // base_types.hpp:66 (24 bytes, same layout as mm_interval_point)
  offset_t pos; hash_t hash; seqno_t seqId; side_t side;
  bool operator<(const IntervalPoint& x) const {
    if (seqId != x.seqId) return seqId < x.seqId;
    if (pos != x.pos) return pos < x.pos;
    return side < x.side;
  }
};
typedef hash_t MinmerMapKeyType;
typedef std::vector<IntervalPoint> MinmerMapValueType;
struct ContigInfo { std::string name; offset_t len; };          // base_types.hpp:94

enum strnd : strand_t { FWD = 1, AMBIG = 0, REV = -1 };
enum event : int { BEGIN = 1, END = 2 };
enum filter : int { MAP = 1, ONETOONE = 2, NONE = 3 };
enum side : side_t { OPEN = 1, CLOSE = -1 };

struct MappingResult {         // base_types.hpp:154
  offset_t queryLen, refStartPos, refEndPos, queryStartPos, queryEndPos;
  seqno_t refSeqId, querySeqId;
  int blockLength;
  float nucIdentity, nucIdentityUpperBound;
  int sketchSize, conservedSketches;
  strand_t strand;
  int approxMatches;
  long double kmerComplexity;
  int n_merged;
  offset_t splitMappingId;
  uint8_t discard;
  bool selfMapFilter;

  size_t hash() {              // base_types.hpp:188 (used by --sparsifyMappings)
    size_t s = 0;
    auto mix = [&s](size_t hv) { s ^= hv + 0x9e3779b9 + (s << 6) + (s >> 2); };
    mix(std::hash<offset_t>()(queryLen)); mix(std::hash<offset_t>()(refStartPos)); mix(std::hash<offset_t>()(refEndPos));
    mix(std::hash<offset_t>()(queryStartPos)); mix(std::hash<offset_t>()(queryEndPos)); mix(std::hash<seqno_t>()(refSeqId));
    mix(std::hash<seqno_t>()(querySeqId)); mix(std::hash<int>()(blockLength)); mix(std::hash<float>()(nucIdentity));
    mix(std::hash<float>()(nucIdentityUpperBound)); mix(std::hash<int>()(sketchSize)); mix(std::hash<int>()(conservedSketches));
    mix(std::hash<strand_t>()(strand)); mix(std::hash<int>()(approxMatches));
    return s;
  }
};
typedef std::vector<MappingResult> MappingResultsVector_t;

struct Parameters {            // map_parameters.hpp:32
  int kmerSize = 19;
  float kmer_pct_threshold = 0.001f;
  offset_t segLength = 5000;
  offset_t block_length = 5000;
  offset_t chain_gap = 5000;
  int alphabetSize = 4;
  offset_t referenceSize = 0;
  float percentageIdentity = 0.85f;
  bool stage2_full_scan = true;
  bool stage1_topANI_filter = true;
  float ANIDiff = 0.0f;
  float ANIDiffConf = 0.999f;
  int filterMode = filter::MAP;
  uint32_t numMappingsForSegment = 1;
  uint32_t numMappingsForShortSequence = 1;
  int threads = 1;
  std::vector<std::string> refSequences;
  std::vector<std::string> querySequences;
  std::string outFileName = "mashmap.out";
  std::filesystem::path saveIndexFilename;
  std::filesystem::path loadIndexFilename;
  bool split = true;
  bool lower_triangular = false;
  bool skip_self = false;
  bool skip_prefix = false;
  char prefix_delim = '\0';
  std::string target_list;
  std::string target_prefix;
  bool mergeMappings = true;
  bool keep_low_pct_id = true;
  bool report_ANI_percentage = false;
  bool filterLengthMismatches = false;
  float kmerComplexityThreshold = 0.0f;
  int sketchSize = 0;
  uint64_t sparsity_hash_threshold = std::numeric_limits<uint64_t>::max();
  bool legacy_output = false;
};

namespace fixed {              // map_parameters.hpp:86
static const double ss_table_max = 1000.0;
static const double pval_cutoff = 1e-3;
static const float confidence_interval = 0.95f;
static const float percentage_identity = 0.85f;
static const float ANIDiff = 0.0f;
static const float ANIDiffConf = 0.999f;
static const std::string VERSION = "3.1.3";
}

}  // namespace skch
#endif

// Page-locked batch buffers for the query reader (skch::Map).  Locking pages costs about a second per few GB, so the buffers are
// allocated by a background thread while skch::Sketch builds the reference index, and recycled between batches afterwards.  Their
// size and number follow the query files (queryBatchPlan): a small job locks one small buffer, not gigabytes; the background thread
// stops as soon as nobody will ask any more (stop(): skch::Map when mapping has finished, and the destructor).
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>
extern "C" { void* mm_host_alloc(size_t bytes); void mm_host_free(void* p); }
namespace skch {
class HostBufferPool {
  std::mutex mu_; std::condition_variable cv_; std::vector<std::pair<char*, size_t>> free_; std::thread bg_; size_t pending_ = 0; std::atomic<bool> stop_{false};
 public:
  static HostBufferPool& instance() { static HostBufferPool p; return p; }
  ~HostBufferPool() { stop(); for (auto& b : free_) mm_host_free(b.first); }
  // no further buffers will be asked for: the background thread ends after the allocation it is in
  void stop() { stop_ = true; if (bg_.joinable()) bg_.join(); std::lock_guard<std::mutex> lk(mu_); pending_ = 0; cv_.notify_all(); }
  void prefetch(size_t n, size_t bytes) {
    if (bg_.joinable()) bg_.join();
    stop_ = false;
    { std::lock_guard<std::mutex> lk(mu_); pending_ = n; }
    bg_ = std::thread([this, n, bytes]() {
      for (size_t i = 0; i < n; i++) {
        if (stop_) { std::lock_guard<std::mutex> lk(mu_); pending_ = 0; cv_.notify_all(); return; }
        char* p = (char*)mm_host_alloc(bytes);
        std::lock_guard<std::mutex> lk(mu_);
        if (p) free_.emplace_back(p, bytes);
        pending_--; cv_.notify_all();
      }
    });
  }
  // a buffer of at least `bytes` if one is ready (or about to be); {nullptr, 0} when the caller should allocate itself
  std::pair<char*, size_t> take(size_t bytes) {
    std::unique_lock<std::mutex> lk(mu_);
    while (true) {
      for (size_t i = 0; i < free_.size(); i++) if (free_[i].second >= bytes) { auto b = free_[i]; free_.erase(free_.begin() + (std::ptrdiff_t)i); return b; }
      if (!pending_) return {nullptr, 0};
      cv_.wait(lk);
    }
  }
  void give(char* p, size_t bytes) { if (!p) return; std::lock_guard<std::mutex> lk(mu_); free_.emplace_back(p, bytes); cv_.notify_all(); }
};

// Page-locked, uninitialised storage for the records a batch brings back from the GPU (48-byte candidate mappings).  A std::vector
// would zero megabytes per batch before the copy overwrites them, and a copy into pageable memory is staged by the runtime at a fifth
// of the link's rate; the buffers are recycled through a free list, a batch that outgrows every free one page-locks a larger one, and
// plain malloc stands in when page-locking fails.
template <class T>
class PinnedRecs {
  struct Pool { std::mutex mu; std::vector<std::pair<void*, size_t>> free_;
                ~Pool() { for (auto& b : free_) mm_host_free(b.first); } };
  static Pool& pool() { static Pool p; return p; }
  T* p_ = nullptr; size_t n_ = 0, cap_ = 0; bool pinned_ = false;
  void release() {
    if (!p_) return;
    if (pinned_) { std::lock_guard<std::mutex> lk(pool().mu); pool().free_.emplace_back((void*)p_, cap_ * sizeof(T)); }
    else free(p_);
    p_ = nullptr; n_ = cap_ = 0; pinned_ = false;
  }
 public:
  PinnedRecs() = default;
  PinnedRecs(const PinnedRecs&) = delete; PinnedRecs& operator=(const PinnedRecs&) = delete;
  PinnedRecs(PinnedRecs&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), pinned_(o.pinned_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
  PinnedRecs& operator=(PinnedRecs&& o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; pinned_ = o.pinned_; o.p_ = nullptr; o.n_ = o.cap_ = 0; } return *this; }
  ~PinnedRecs() { release(); }
  T* data() { return p_; } const T* data() const { return p_; }
  size_t size() const { return n_; } bool empty() const { return n_ == 0; }
  T& operator[](size_t i) { return p_[i]; } const T& operator[](size_t i) const { return p_[i]; }
  // page-locks `count` buffers of `n` records each ahead of their use (skch::Sketch does it in the background of the index build: the
  // records of a device pass then never wait for pages to be locked)
  static void prefill(size_t count, size_t n) {
    for (size_t i = 0; i < count; i++) {
      const size_t bytes = std::max<size_t>(n * sizeof(T), (size_t)1 << 20);
      void* q = mm_host_alloc(bytes);
      if (!q) return;
      std::lock_guard<std::mutex> lk(pool().mu);
      pool().free_.emplace_back(q, bytes);
    }
  }
  // contents are NOT kept and NOT initialised
  void resize(size_t n) {
    if (n <= cap_) { n_ = n; return; }
    release();
    const size_t want = n * sizeof(T);
    {
      std::lock_guard<std::mutex> lk(pool().mu);
      auto& f = pool().free_;
      for (size_t i = 0; i < f.size(); i++) if (f[i].second >= want) { p_ = (T*)f[i].first; cap_ = f[i].second / sizeof(T); pinned_ = true; f.erase(f.begin() + (std::ptrdiff_t)i); break; }
    }
    if (!p_) {
      const size_t bytes = std::max<size_t>(want + want / 4, (size_t)1 << 20);
      if (void* q = mm_host_alloc(bytes)) { p_ = (T*)q; cap_ = bytes / sizeof(T); pinned_ = true; }
      else { p_ = (T*)malloc(want); cap_ = n; pinned_ = false; if (!p_) { std::fprintf(stderr, "[mashmap_hip] out of memory\n"); exit(1); } }
    }
    n_ = n;
  }
};

// How skch::Map takes the query files through the GPUs.  The reader's unit is a BATCH: MASHMAP_HIP_BATCH_MBP (default 512 Mbp) PER GPU
// CONTEXT, parsed and packed into one page-locked buffer -- small, so that locking its pages is cheap and the three stages overlap from
// the first few milliseconds on.  The device's unit is a PASS: up to MASHMAP_HIP_COALESCE_MBP (default 3072 Mbp per context; 0 = one batch
// per pass) of consecutive batches laid end to end in HBM (mm_reads_upload_packed_parts) -- the kernels of a 512 Mbp pass leave a third
// of the GPU idle (1.5 waves per SIMD in the lane-per-candidate sweep); measured inside the command line a 2 Gbp pass runs at 138-142 Gbp/s
// (every kernel's tail and launch is paid once per pass), a 3 Gbp pass at ~145.  Page-locked buffers: a buffer is
// busy from the reader's first byte until its batch's upload has completed (the post stage works on the records, not on the bases):
// one being parsed + a pass's worth queued + a pass's worth uploading, but no more than the input needs, each no larger than the input.
struct QueryBatchPlan { size_t batchBases; size_t passBases; size_t bufferBytes; size_t buffers; uint64_t inputBytes; bool inputKnown; };
inline QueryBatchPlan queryBatchPlan(const std::vector<std::string>& queryFiles, size_t nContexts) {
  const bool packed = getenv("MASHMAP_HIP_ASCII_UPLOAD") == nullptr;   // the reader packs: a batch buffer holds 3/8 byte per base, not 1
  const char* be = getenv("MASHMAP_HIP_BATCH_MBP");
  const char* ce = getenv("MASHMAP_HIP_COALESCE_MBP");
  const size_t nCtx = nContexts ? nContexts : 1;
  QueryBatchPlan q;
  q.batchBases = (size_t)((be ? atof(be) : 512.0) * 1e6) * nCtx;
  if (q.batchBases < 1) q.batchBases = 1;
  q.passBases = std::max(q.batchBases, (size_t)((ce ? atof(ce) : 3072.0) * 1e6) * nCtx);
  if (nCtx > 1 || !packed) q.passBases = q.batchBases;               // (several batches per pass: one context, packed uploads -- skch_map.hpp)
  q.passBases = std::min(q.passBases, q.batchBases * 64);            // at most 64 batches per pass, whatever the two variables say
  const size_t perPass = q.passBases / q.batchBases;
  q.inputBytes = 0; q.inputKnown = !queryFiles.empty();
  for (const auto& f : queryFiles) {
    struct stat st;
    if (stat(f.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) { q.inputKnown = false; continue; }     // a pipe: size unknown
    bool gz = false;
    if (FILE* fp = fopen(f.c_str(), "rb")) { unsigned char m[2] = {0, 0}; gz = fread(m, 1, 2, fp) == 2 && m[0] == 31 && m[1] == 139; fclose(fp); }
    q.inputBytes += (uint64_t)st.st_size * (gz ? 5u : 1u);           // DNA text deflates to between a fifth and a third
  }
  const size_t ascii = q.batchBases + q.batchBases / 8 + (1u << 20);
  const size_t full = (packed && !getenv("MASHMAP_HIP_BIG_BUFFERS")) ? (ascii + (2u << 20)) / 8 * 3 + (1u << 20) : ascii;      // locking pages costs ~0.2 s per GB: no more than needed
  const size_t inFlight = std::min<size_t>(24, 2 * perPass + 2) + (perPass == 1 ? 4 : 0);   // one batch per pass: as before (reader 1 + two queues of 2 + device 1 + post 1 + one spare)
  if (!q.inputKnown) { q.bufferBytes = full; q.buffers = inFlight; return q; }
  const uint64_t batches = q.inputBytes / q.batchBases + 1;
  const uint64_t whole = q.inputBytes + q.inputBytes / 8 + (1u << 20);
  q.bufferBytes = (size_t)std::min<uint64_t>(full, packed ? whole / 8 * 3 + (2u << 20) : whole);
  q.buffers = (size_t)std::min<uint64_t>(inFlight, batches + 1);
  return q;
}
// packed bases the per-context staging area of the early prefetch holds (skch::Map sends parsed batches ahead into it; skch::Sketch
// allocates it behind the index build): two passes' worth, but no more than the input, plus a quarter for the 32-base alignment of
// every read and the parser's gaps
inline size_t stagingCapBases(const QueryBatchPlan& plan, size_t nContexts) {
  const size_t cap = 2 * std::max(plan.passBases, plan.batchBases) / std::max<size_t>(1, nContexts);
  return plan.inputKnown ? (size_t)std::min<uint64_t>((uint64_t)cap, plan.inputBytes + (64u << 20)) : cap;
}
inline size_t stagingReserveBases(const QueryBatchPlan& plan, size_t nContexts) {
  const size_t cap = stagingCapBases(plan, nContexts);
  return cap + cap / 4 + (1u << 22);
}
}  // namespace skch

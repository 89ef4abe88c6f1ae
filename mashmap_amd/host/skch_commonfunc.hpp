// mashmap_amd/host/skch_commonfunc.hpp -- skch::CommonFunc::sketchSequence / addMinmers on top of the C ABI (include/mashmap_hip.h).
//
// The inner seams of the reference's hot path: code written against the commonFunc.hpp-level API -- wfmash-style
// callers, test harnesses that sketch one sequence at a time -- keeps compiling, with the same template signatures
//   sketchSequence(std::vector<T>&, char* seq, offset_t len, int kmerSize, int alphabetSize, int sketchSize, seqno_t seqCounter)
//                                                                                   src/map/include/commonFunc.hpp:183-288
//   addMinmers(std::vector<T>&, char* seq, offset_t len, int kmerSize, int windowSize, int alphabetSize, int sketchSize, seqno_t)
//                                                                                   src/map/include/commonFunc.hpp:302-570
// and the same observable effects: `seq` is normalised in place (makeUpperCaseAndValidDNA, :97), sketchSequence REPLACES the
// vector's content with the sketch (:278-286), addMinmers APPENDS the contig's records (:514, :555; the reference hands it a fresh
// vector per contig, winSketch.hpp:241-252).  The work is done by the kernels behind mm_sketch_fragments / mm_index_build on batches
// of one fragment / one contig -- a compatibility path, not a fast one: callers that have many sequences should use skch::Sketch /
// skch::Map (or the C ABI) with whole batches.  There is no CPU fallback: without a gfx950 device the calls abort with the library's
// error text.  Only alphabetSize == 4 (DNA) exists on the device.
//
// Inside the reference tree (-DMASHMAP_HIP_REFERENCE_TREE, reference_tree/map/include/commonFunc.hpp) the reference's own header is
// still included for everything else it defines (getHash, reverseComplement, split, getReferenceSize ...); only these two templates
// are replaced.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/mashmap_hip.h"
#include "skch_types.hpp"

namespace skch {
namespace hipseam {

// one context per (device, k, segment length, sketch size), created on first use.  A caller that sketches sequences of many different
// lengths (the segment length is the sequence length rounded up) would otherwise pile up contexts -- each holds device buffers -- so the
// cache keeps the MM_SEAM_CONTEXTS (8) most recently used and destroys the rest.  Callers hold callMutex() while they use a context,
// which is also what makes the eviction safe.
#ifndef MM_SEAM_CONTEXTS
#define MM_SEAM_CONTEXTS 8
#endif
inline mm_ctx* context(int kmerSize, int segLength, int sketchSize) {
  struct Entry { mm_ctx* ctx; uint64_t lastUse; };
  static std::map<std::tuple<int, int, int, int>, Entry> cache;
  static uint64_t tick = 0;
  const char* de = getenv("MASHMAP_HIP_DEVICE");
  const int dev = de ? atoi(de) : 0;
  auto key = std::make_tuple(dev, kmerSize, segLength, sketchSize);
  auto it = cache.find(key);
  if (it != cache.end()) { it->second.lastUse = ++tick; return it->second.ctx; }
  if (cache.size() >= MM_SEAM_CONTEXTS) {                       // least recently used out
    auto victim = cache.begin();
    for (auto j = cache.begin(); j != cache.end(); ++j) if (j->second.lastUse < victim->second.lastUse) victim = j;
    mm_destroy(victim->second.ctx);
    cache.erase(victim);
  }
  mm_params p; p.kmerSize = kmerSize; p.segLength = segLength; p.sketchSize = sketchSize; p.flags = MM_FLAG_NO_SPLIT;
  mm_ctx* c = nullptr;
  if (mm_create(&c, dev, &p) != MM_OK) {
    std::cerr << "[mashmap_hip::skch::CommonFunc] ERROR: " << mm_last_error(nullptr) << std::endl;
    exit(1);
  }
  cache.emplace(key, Entry{c, ++tick});
  return c;
}
[[noreturn]] inline void die(const char* what, mm_ctx* c) {
  std::cerr << "[mashmap_hip::skch::CommonFunc] ERROR: " << what << ": " << mm_last_error(c) << std::endl;
  exit(1);
}
inline std::mutex& callMutex() { static std::mutex m; return m; }      // an mm_ctx is thread-compatible, not thread-safe

// makeUpperCaseAndValidDNA (commonFunc.hpp:97-107): a-z -> A-Z, everything but A C G T -> N.  (The reference indexes its 127-entry
// table with a signed char; bytes >= 127 are out of its bounds there -- here they become N, which is what the device does too.)
inline void normalise(char* seq, offset_t len) {
  for (offset_t i = 0; i < len; i++) {
    unsigned char ch = (unsigned char)seq[i];
    if (ch > 96 && ch < 123) ch -= 32;
    seq[i] = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? (char)ch : 'N';
  }
}

template <typename T>
inline void sketchSequence(std::vector<T>& minmerIndex, char* seq, offset_t len, int kmerSize, int alphabetSize, int sketchSize, seqno_t seqCounter) {
  if (alphabetSize != 4) { std::cerr << "[mashmap_hip::skch::CommonFunc] ERROR: only alphabetSize 4 (DNA) runs on the device" << std::endl; exit(1); }
  normalise(seq, len);
  minmerIndex.clear();
  if (len < kmerSize) return;                                           // no k-mer: the loop of :218 never runs
  // the fragment is sketched whole (MM_FLAG_NO_SPLIT context whose segLength is the next multiple of 1024 above it)
  const int L = (int)((((int64_t)len + 1023) / 1024) * 1024);
  std::lock_guard<std::mutex> lk(callMutex());
  mm_ctx* c = context(kmerSize, L, sketchSize);
  const int64_t offs[2] = {0, (int64_t)len};
  if (mm_reads_upload(c, seq, offs, 1, nullptr, nullptr, seqCounter) != MM_OK) die("mm_reads_upload", c);
  if (mm_num_fragments(c) != 1) die("one read, one fragment expected", c);
  if (mm_sketch_fragments(c) != MM_OK) die("mm_sketch_fragments", c);
  std::vector<mm_minmer> out((size_t)sketchSize);
  uint32_t n = 0;
  if (mm_sketch_download(c, out.data(), &n) != MM_OK) die("mm_sketch_download", c);
  minmerIndex.resize(n);
  for (uint32_t i = 0; i < n; i++) minmerIndex[i] = T{out[i].hash, out[i].wpos, out[i].wpos_end, out[i].seqId, (strand_t)out[i].strand};
}

template <typename T>
inline void addMinmers(std::vector<T>& minmerIndex, char* seq, offset_t len, int kmerSize, int windowSize, int alphabetSize, int sketchSize, seqno_t seqCounter) {
  if (alphabetSize != 4) { std::cerr << "[mashmap_hip::skch::CommonFunc] ERROR: only alphabetSize 4 (DNA) runs on the device" << std::endl; exit(1); }
  normalise(seq, len);
  if (len < windowSize || len < kmerSize) return;                      // no complete window: nothing is emitted (:341, :455)
  std::lock_guard<std::mutex> lk(callMutex());
  mm_ctx* c = context(kmerSize, windowSize, sketchSize);
  if (mm_set_option(c, MM_OPT_KEEP_FULL_INDEX, 1) != MM_OK) die("mm_set_option", c);   // the records BEFORE the frequent-seed drop are addMinmers' output
  const int64_t offs[2] = {0, (int64_t)len};
  if (mm_index_build(c, seq, offs, 1, nullptr, 0.0f) != MM_OK) die("mm_index_build", c);
  size_t n = 0;
  if (mm_index_download_full(c, nullptr, &n) != MM_OK) die("mm_index_download_full", c);
  std::vector<mm_minmer> out(n);
  if (n && mm_index_download_full(c, out.data(), &n) != MM_OK) die("mm_index_download_full", c);
  minmerIndex.reserve(minmerIndex.size() + n);
  for (size_t i = 0; i < n; i++) minmerIndex.push_back(T{out[i].hash, out[i].wpos, out[i].wpos_end, seqCounter, (strand_t)out[i].strand});
}

}  // namespace hipseam

#ifndef MASHMAP_HIP_REFERENCE_TREE
namespace CommonFunc {
inline void makeUpperCaseAndValidDNA(char* seq, offset_t len) { hipseam::normalise(seq, len); }
template <typename T>
inline void sketchSequence(std::vector<T>& minmerIndex, char* seq, offset_t len, int kmerSize, int alphabetSize, int sketchSize, seqno_t seqCounter) {
  hipseam::sketchSequence(minmerIndex, seq, len, kmerSize, alphabetSize, sketchSize, seqCounter);
}
template <typename T>
inline void addMinmers(std::vector<T>& minmerIndex, char* seq, offset_t len, int kmerSize, int windowSize, int alphabetSize, int sketchSize, seqno_t seqCounter) {
  hipseam::addMinmers(minmerIndex, seq, len, kmerSize, windowSize, alphabetSize, sketchSize, seqCounter);
}
}  // namespace CommonFunc
#endif

}  // namespace skch

// mashmap_amd/host/skch_sketch.hpp -- skch::Sketch on top of the C ABI (include/mashmap_hip.h).
//
// Stands in for the reference's class of the same name (src/map/include/winSketch.hpp:57-513): same constructor
// (builds + indexes the reference in the ctor, :122-138), same public members (metadata :79, sequencesByFileInfo :88,
// minmerPosLookupIndex :101, minmerIndex :102) and accessors (isFreqSeed :506, getFreqThreshold :483,
// isMinmerIndexEnd / getMinmerIndexEnd :470-481).  What differs is where the work happens: the FASTA text is read on the
// host (seq_reader.hpp), everything from the k-mer hashes on is done by mm_index_build (a5-a7), and the index that
// Map reads lives in HBM behind the mm_ctx this object owns.  The host copies of minmerIndex / minmerPosLookupIndex
// are materialised from the library (mm_index_download) -- minmerIndex eagerly (one memcpy), the hash map only on
// request (materializeLookupIndex), because nothing in the device path reads it.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/mashmap_hip.h"
#include "seq_reader.hpp"
#include "skch_types.hpp"

namespace skch {

class Sketch {
  const skch::Parameters& param;
  int freqThreshold = std::numeric_limits<int>::max();
  std::vector<hash_t> frequentSeeds;               // ascending
  mm_ctx* ctx_ = nullptr;
  Sketch();

  [[noreturn]] void die(const char* what) const {
    std::cerr << "[mashmap_hip::skch::Sketch] ERROR: " << what << ": " << mm_last_error(ctx_) << std::endl;
    exit(1);
  }

 public:
  typedef std::vector<MinmerInfo> MI_Type;
  using MIIter_t = MI_Type::const_iterator;
  using MI_Map_t = std::unordered_map<MinmerMapKeyType, MinmerMapValueType>;

  std::vector<ContigInfo> metadata;
  std::vector<int> sequencesByFileInfo;
  MI_Map_t minmerPosLookupIndex;                   // see materializeLookupIndex()
  MI_Type minmerIndex;

  static int deviceFromEnv() { const char* e = getenv("MASHMAP_HIP_DEVICE"); return e ? atoi(e) : 0; }

  explicit Sketch(const skch::Parameters& p) : param(p) {
    static_assert(sizeof(MinmerInfo) == sizeof(mm_minmer), "MinmerInfo layout");
    static_assert(sizeof(IntervalPoint) == sizeof(mm_interval_point), "IntervalPoint layout");
    if (!p.saveIndexFilename.empty() || !p.loadIndexFilename.empty()) {
      std::cerr << "[mashmap_hip::skch::Sketch] ERROR: --saveIndex/--loadIndex are not supported by the device index (SURVEY 8f.4)" << std::endl;
      exit(1);
    }
    mm_params mp;
    mp.kmerSize = p.kmerSize; mp.segLength = p.segLength; mp.sketchSize = p.sketchSize;
    mp.flags = (p.stage1_topANI_filter ? MM_FLAG_HG_FILTER : 0) | (p.skip_self ? MM_FLAG_SKIP_SELF : 0) |
               (p.skip_prefix ? MM_FLAG_SKIP_PREFIX : 0) | (p.lower_triangular ? MM_FLAG_LOWER_TRIANGULAR : 0) |
               (p.split ? 0 : MM_FLAG_NO_SPLIT);
    if (mm_create(&ctx_, deviceFromEnv(), &mp) != MM_OK) {
      std::cerr << "[mashmap_hip::skch::Sketch] ERROR: " << mm_last_error(nullptr) << std::endl;
      exit(1);
    }
    this->build();
  }
  ~Sketch() { mm_destroy(ctx_); }
  Sketch(const Sketch&) = delete;
  Sketch& operator=(const Sketch&) = delete;

  mm_ctx* ctx() const { return ctx_; }

  // Map::setRefGroups (computeMap.hpp:144): consecutive contigs with equal name prefix share a group
  std::vector<int> refGroups() const {
    std::vector<int> g(metadata.size(), 0);
    int group = 0; size_t start = 0;
    auto prefix = [this](const std::string& s) { return s.substr(0, s.find_last_of(param.prefix_delim)); };
    while (start < metadata.size()) {
      const std::string cur = prefix(metadata[start].name);
      size_t i = start;
      while (i < metadata.size() && prefix(metadata[i].name) == cur) g[i++] = group;
      group++; start = i;
    }
    return g;
  }

  int getFreqThreshold() const { return freqThreshold; }
  bool isFreqSeed(hash_t h) const { return std::binary_search(frequentSeeds.begin(), frequentSeeds.end(), h); }
  MIIter_t searchIndex(seqno_t seqId, offset_t winpos) const {
    return std::lower_bound(minmerIndex.begin(), minmerIndex.end(), MinmerInfo{0, winpos, 0, seqId, 0});
  }
  bool isMinmerIndexEnd(const MIIter_t& it) const { return it == minmerIndex.end(); }
  MIIter_t getMinmerIndexEnd() const { return minmerIndex.end(); }

  // the hash -> interval points map of winSketch.hpp:100-101 on the host (only callers outside the device path need it)
  void materializeLookupIndex() {
    size_t nM, nK, nP, nF; int32_t ft;
    if (mm_index_sizes(ctx_, &nM, &nK, &nP, &nF, &ft) != MM_OK) die("mm_index_sizes");
    std::vector<uint64_t> keys(nK), offs(nK + 1);
    std::vector<mm_interval_point> pts(nP);
    if (mm_index_download(ctx_, nullptr, keys.data(), offs.data(), pts.data(), nullptr) != MM_OK) die("mm_index_download");
    minmerPosLookupIndex.clear(); minmerPosLookupIndex.reserve(nK);
    for (size_t i = 0; i < nK; i++) {
      auto& v = minmerPosLookupIndex[keys[i]];
      for (uint64_t j = offs[i]; j < offs[i + 1]; j++) v.push_back(IntervalPoint{pts[j].pos, pts[j].hash, pts[j].seqId, pts[j].side});
    }
  }

 private:
  void build() {      // winSketch.hpp:147-231 + :379-504, with the compute moved behind mm_index_build
    std::unordered_set<std::string> allowed;
    if (!param.target_list.empty()) {
      std::ifstream fl(param.target_list);
      std::string name;
      while (std::getline(fl, name)) allowed.insert(name);
    }
    std::string bases;
    std::vector<int64_t> offs(1, 0);
    seqno_t seqCounter = 0;
    for (const auto& fileName : param.refSequences) {
      mmhost::for_each_seq_in_file(fileName, allowed, param.target_prefix, [&](const std::string& name, std::string& seq) {
        metadata.push_back(ContigInfo{name, (offset_t)seq.length()});
        bases.append(seq);
        offs.push_back((int64_t)bases.size());
        seqCounter++;
      });
      sequencesByFileInfo.push_back(seqCounter);
    }
    if (seqCounter == 0) {
      std::cerr << "[mashmap::skch::Sketch::build] ERROR: No sequences indexed!" << std::endl;
      exit(1);
    }
    std::vector<int> groups;
    if (param.skip_prefix) groups = refGroups();
    if (mm_index_build(ctx_, bases.data(), offs.data(), metadata.size(), param.skip_prefix ? groups.data() : nullptr,
                       param.kmer_pct_threshold) != MM_OK) die("mm_index_build");
    size_t nM, nK, nP, nF; int32_t ft;
    if (mm_index_sizes(ctx_, &nM, &nK, &nP, &nF, &ft) != MM_OK) die("mm_index_sizes");
    minmerIndex.resize(nM);
    frequentSeeds.resize(nF);
    if (mm_index_download(ctx_, reinterpret_cast<mm_minmer*>(minmerIndex.data()), nullptr, nullptr, nullptr, frequentSeeds.data()) != MM_OK)
      die("mm_index_download");
    std::sort(frequentSeeds.begin(), frequentSeeds.end());
    freqThreshold = ft;
    std::cerr << "[mashmap::skch::Sketch::build] minmer windows picked from reference (after frequent-seed removal) = " << nM << std::endl;
    std::cerr << "[mashmap::skch::Sketch::index] unique minmers = " << nK << std::endl;
    if (nF == 0) std::cerr << "[mashmap::skch::Sketch::computeFreqHist] With threshold " << param.kmer_pct_threshold
                           << "%, consider all minmers during lookup." << std::endl;
    else std::cerr << "[mashmap::skch::Sketch::computeFreqHist] With threshold " << param.kmer_pct_threshold << "%, ignore minmers with more than >= "
                   << freqThreshold << " interval points during mapping." << std::endl;
  }
};

}  // namespace skch

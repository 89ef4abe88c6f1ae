// mashmap_amd/host/skch_sketch.hpp -- skch::Sketch on top of the C ABI (include/mashmap_hip.h).
//
// Stands in for the reference's class of the same name (src/map/include/winSketch.hpp:57-513): same constructor
// (builds + indexes the reference in the ctor, :122-138), same public members (metadata :79, sequencesByFileInfo :88,
// minmerPosLookupIndex :101, minmerIndex :102) and accessors (isFreqSeed :506, getFreqThreshold :483,
// isMinmerIndexEnd / getMinmerIndexEnd :470-481).  What differs is where the work happens: the FASTA text is read on the
// host (seq_parse.hpp, multi-threaded), everything from the k-mer hashes on is done by mm_index_build (a5-a7), and the index that
// Map reads lives in HBM behind the mm_ctx this object owns.  The host copies of minmerIndex / minmerPosLookupIndex
// are materialised from the library (mm_index_download) -- minmerIndex eagerly (one memcpy), the hash map only on
// request (materializeLookupIndex), because nothing in the device path reads it.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/mashmap_hip.h"
#include "seq_parse.hpp"
#include "skch_types.hpp"

namespace skch {

class Sketch {
  const skch::Parameters& param;
  int freqThreshold = std::numeric_limits<int>::max();
  std::vector<hash_t> frequentSeeds;               // ascending
  size_t nMinmers_ = 0;                            // |minmerIndex| on the device (after the frequent-seed drop)
  mm_ctx* ctx_ = nullptr;                          // the context the index is built on (first device of the list)
  std::vector<mm_ctx*> ctxs_;                      // one per entry of MASHMAP_HIP_DEVICES; ctxs_[0] == ctx_
  bool commReady_ = false;
  mutable std::mutex materializeMu_; mutable bool minmerIndexReady_ = false;
  std::thread recsPrefill_;                        // page-locks the record buffers of skch::Map's device passes while the index is built
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
  mutable MI_Type minmerIndex;                     // filled on first use (materializeMinmerIndex)

  // MASHMAP_HIP_DEVICES=0,1,... : the GPUs query batches are sharded over (index replicated);
  // MASHMAP_HIP_DEVICE=n : a single one (default 0).  A device may be listed twice (two contexts on one GPU).
  static std::vector<int> devicesFromEnv() {
    std::vector<int> d;
    if (const char* e = getenv("MASHMAP_HIP_DEVICES")) {
      std::stringstream ss(e); std::string tok;
      while (std::getline(ss, tok, ',')) if (!tok.empty()) d.push_back(atoi(tok.c_str()));
    }
    if (d.empty()) { const char* e = getenv("MASHMAP_HIP_DEVICE"); d.push_back(e ? atoi(e) : 0); }
    return d;
  }

  explicit Sketch(const skch::Parameters& p) : param(p) {
    static_assert(sizeof(MinmerInfo) == sizeof(mm_minmer), "MinmerInfo layout");
    static_assert(sizeof(IntervalPoint) == sizeof(mm_interval_point), "IntervalPoint layout");
    mm_params mp;
    mp.kmerSize = p.kmerSize; mp.segLength = p.segLength; mp.sketchSize = p.sketchSize;
    mp.flags = (p.stage1_topANI_filter ? MM_FLAG_HG_FILTER : 0) | (p.skip_self ? MM_FLAG_SKIP_SELF : 0) |
               (p.skip_prefix ? MM_FLAG_SKIP_PREFIX : 0) | (p.lower_triangular ? MM_FLAG_LOWER_TRIANGULAR : 0) |
               (p.split ? 0 : MM_FLAG_NO_SPLIT);
    for (int dev : devicesFromEnv()) {
      mm_ctx* c = nullptr;
      if (mm_create(&c, dev, &mp) != MM_OK) {
        std::cerr << "[mashmap_hip::skch::Sketch] ERROR: " << mm_last_error(nullptr) << std::endl;
        exit(1);
      }
      ctxs_.push_back(c);
    }
    ctx_ = ctxs_[0];
    {
      // page-locked buffers for the query batches skch::Map will read: locked in the background while the index is built; sized and
      // counted by what the query files hold (skch_types.hpp: queryBatchPlan)
      const QueryBatchPlan plan = queryBatchPlan(p.querySequences, ctxs_.size());
      HostBufferPool::instance().prefetch(plan.buffers, plan.bufferBytes);
      // ... and so are the buffers the candidate mappings of a device pass come back into (about one 48-byte record per segment; three
      // passes can hold one at a time: device, queue, post stage)
      if (!plan.inputKnown || plan.inputBytes > (64u << 20)) {
        const size_t perPass = (size_t)(std::min<uint64_t>(plan.passBases, plan.inputKnown ? plan.inputBytes : plan.passBases) / (uint64_t)std::max<offset_t>(1, p.segLength));
        recsPrefill_ = std::thread([perPass]() { PinnedRecs<mm_mapping>::prefill(6, perPass + perPass / 4 + 1024); });
      }
      // ... and the query files' pages are mapped into this process meanwhile (seq_parse.hpp: MappedFileCache)
      if (!getenv("MASHMAP_HIP_NO_PREFAULT")) mmhost::MappedFileCache::instance().prefault(p.querySequences, 8);
    }
    if (!p.saveIndexFilename.empty()) mm_set_option(ctx_, MM_OPT_KEEP_FULL_INDEX, 1);
    this->build();
    if (!p.saveIndexFilename.empty()) this->saveIndex();
    for (size_t i = 1; i < ctxs_.size(); i++)        // replicas of the resident index, GPU to GPU
      if (mm_index_replicate(ctxs_[i], ctx_) != MM_OK) { std::cerr << "[mashmap_hip::skch::Sketch] ERROR: mm_index_replicate: " << mm_last_error(ctxs_[i]) << std::endl; exit(1); }
    if (!getenv("MASHMAP_HIP_ASCII_UPLOAD") && !getenv("MASHMAP_HIP_NO_EARLY_PREFETCH")) {
      // the staging area skch::Map's reader sends its packed batches ahead into (skch_map.hpp: issuePrefetch reserves the same size): allocated
      // here, behind the index build, so that the reader's second batch does not wait ~10 ms for a multi-gigabyte hipMalloc
      const QueryBatchPlan plan = queryBatchPlan(p.querySequences, ctxs_.size());
      for (mm_ctx* c : ctxs_) (void)mm_reads_prefetch_reserve(c, stagingReserveBases(plan, ctxs_.size()));
    }
    if (ctxs_.size() > 1) {
      // the communicator of the device-side exchange step (RCCL between distinct GPUs, peer copies between contexts that share one).  A
      // machine where it cannot be had still maps: skch::Map then lets every context hand its block to the host (one warning here)
      commReady_ = mm_comm_init_local(ctxs_.data(), (int)ctxs_.size()) == MM_OK;
      if (!commReady_) std::cerr << "[mashmap_hip::skch::Sketch] WARNING: mm_comm_init_local: " << mm_last_error(ctx_)
                                 << "; the candidate mappings of the GPUs are exchanged through the host" << std::endl;
    }
  }
  ~Sketch() { if (recsPrefill_.joinable()) recsPrefill_.join(); for (mm_ctx* c : ctxs_) mm_destroy(c); }
  Sketch(const Sketch&) = delete;
  Sketch& operator=(const Sketch&) = delete;

  mm_ctx* ctx() const { return ctx_; }
  const std::vector<mm_ctx*>& contexts() const { return ctxs_; }
  bool commReady() const { return commReady_; }    // mm_comm_init_local succeeded (several contexts only)
  // every context has a GPU of its own: what an RCCL communicator over them needs (two contexts on one GPU exchange by device copies)
  bool distinctDevices() const {
    std::vector<int> d = devicesFromEnv();
    std::sort(d.begin(), d.end());
    return d.size() > 1 && std::adjacent_find(d.begin(), d.end()) == d.end();
  }

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
  // minmerIndex (winSketch.hpp:102) lives on the device; the host copy in reference layout is made on first use
  // (the reference's callers read the index from many threads at once -- mapModule through searchIndex -- so the first use is serialised;
  // code that reads the public member directly calls this first, or sets MASHMAP_HIP_EAGER_INDEX=1)
  void materializeMinmerIndex() const {
    std::lock_guard<std::mutex> lk(materializeMu_);
    if (minmerIndexReady_) return;
    minmerIndex.resize(nMinmers_);
    if (nMinmers_ && mm_index_download(ctx_, reinterpret_cast<mm_minmer*>(minmerIndex.data()), nullptr, nullptr, nullptr, nullptr) != MM_OK) die("mm_index_download");
    minmerIndexReady_ = true;
  }
  size_t minmerIndexSize() const { return nMinmers_; }
  MIIter_t searchIndex(seqno_t seqId, offset_t winpos) const {
    materializeMinmerIndex();
    return std::lower_bound(minmerIndex.begin(), minmerIndex.end(), MinmerInfo{0, winpos, 0, seqId, 0});
  }
  bool isMinmerIndexEnd(const MIIter_t& it) const { return it == minmerIndex.end(); }
  MIIter_t getMinmerIndexEnd() const { materializeMinmerIndex(); return minmerIndex.end(); }

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
  // --saveIndex PREFIX (winSketch.hpp:127-134, 270-315): PREFIX.index = size_t n + raw MinmerInfo[n] of minmerIndex before the
  // frequent-seed drop (or a TSV when PREFIX ends in .tsv), PREFIX.map = size_t keys, then per key: hash, size_t n, raw IntervalPoint[n].
  // Keys are written in order of first appearance in minmerIndex, the iteration order of the reference's insertion-ordered map.
  void saveIndex() {
    size_t nAll = 0;
    if (mm_index_download_full(ctx_, nullptr, &nAll) != MM_OK) die("mm_index_download_full");
    std::vector<MinmerInfo> all(nAll);
    if (mm_index_download_full(ctx_, reinterpret_cast<mm_minmer*>(all.data()), &nAll) != MM_OK) die("mm_index_download_full");
    if (param.saveIndexFilename.extension() == ".tsv") {
      std::ofstream out(param.saveIndexFilename);
      out << "seqId" << "\t" << "strand" << "\t" << "start" << "\t" << "end" << "\t" << "hash\n";
      for (const auto& mi : all) out << mi.seqId << "\t" << std::to_string(mi.strand) << "\t" << mi.wpos << "\t" << mi.wpos_end << "\t" << mi.hash << "\n";
    } else {
      std::filesystem::path fn = param.saveIndexFilename; fn += ".index";
      std::ofstream out(fn, std::ios::binary);
      const size_t n = all.size();
      out.write((const char*)&n, sizeof n);
      out.write((const char*)all.data(), (std::streamsize)(n * sizeof(MinmerInfo)));
    }
    size_t nM, nK, nP, nF; int32_t ft;
    if (mm_index_sizes(ctx_, &nM, &nK, &nP, &nF, &ft) != MM_OK) die("mm_index_sizes");
    std::vector<uint64_t> keys(nK), offs(nK + 1);
    std::vector<mm_interval_point> pts(nP);
    if (mm_index_download(ctx_, nullptr, keys.data(), offs.data(), pts.data(), nullptr) != MM_OK) die("mm_index_download");
    std::vector<uint32_t> first(nK, 0xFFFFFFFFu), order;
    order.reserve(nK);
    for (const auto& mi : all) {
      const size_t ki = (size_t)(std::lower_bound(keys.begin(), keys.end(), mi.hash) - keys.begin());
      if (ki < nK && keys[ki] == mi.hash && first[ki] == 0xFFFFFFFFu) { first[ki] = 1; order.push_back((uint32_t)ki); }
    }
    std::filesystem::path fn = param.saveIndexFilename; fn += ".map";
    std::ofstream out(fn, std::ios::binary);
    const size_t nKeys = order.size();
    out.write((const char*)&nKeys, sizeof nKeys);
    std::vector<IntervalPoint> tmp;
    for (uint32_t ki : order) {
      const MinmerMapKeyType key = keys[ki];
      const size_t n = (size_t)(offs[ki + 1] - offs[ki]);
      tmp.clear();
      for (uint64_t j = offs[ki]; j < offs[ki + 1]; j++) tmp.push_back(IntervalPoint{pts[j].pos, pts[j].hash, pts[j].seqId, pts[j].side});
      out.write((const char*)&key, sizeof key);
      out.write((const char*)&n, sizeof n);
      out.write((const char*)tmp.data(), (std::streamsize)(n * sizeof(IntervalPoint)));
    }
  }

  // --loadIndex PREFIX (winSketch.hpp:166-172, 320-374): minmerIndex and the lookup map from disk; the frequent-seed steps run
  // afterwards, as in the reference's constructor (:135-137), inside mm_index_upload_full
  void loadIndex(const std::vector<int>& groups) {
    std::vector<MinmerInfo> all;
    if (param.loadIndexFilename.extension() == ".tsv") {
      std::ifstream in(param.loadIndexFilename);
      std::string line;
      std::getline(in, line);                                     // header
      while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::stringstream ls(line);
        long long seqId, strand, start, end; unsigned long long hash;
        ls >> seqId >> strand >> start >> end >> hash;
        all.push_back(MinmerInfo{(hash_t)hash, (offset_t)start, (offset_t)end, (seqno_t)seqId, (strand_t)strand});
      }
    } else {
      std::filesystem::path fn = param.loadIndexFilename; fn += ".index";
      std::ifstream in(fn, std::ios::binary);
      size_t n = 0;
      in.read((char*)&n, sizeof n);
      if (!in) { std::cerr << "[mashmap_hip::skch::Sketch] ERROR: cannot read " << fn << std::endl; exit(1); }
      all.resize(n);
      in.read((char*)all.data(), (std::streamsize)(n * sizeof(MinmerInfo)));
    }
    std::filesystem::path fn = param.loadIndexFilename; fn += ".map";
    std::ifstream in(fn, std::ios::binary);
    size_t nKeys = 0;
    in.read((char*)&nKeys, sizeof nKeys);
    if (!in) { std::cerr << "[mashmap_hip::skch::Sketch] ERROR: cannot read " << fn << std::endl; exit(1); }
    std::vector<uint64_t> keys(nKeys), offs(nKeys + 1, 0);
    std::vector<mm_interval_point> pts;
    std::vector<IntervalPoint> tmp;
    for (size_t i = 0; i < nKeys; i++) {
      MinmerMapKeyType key = 0; size_t n = 0;
      in.read((char*)&key, sizeof key);
      in.read((char*)&n, sizeof n);
      tmp.resize(n);
      in.read((char*)tmp.data(), (std::streamsize)(n * sizeof(IntervalPoint)));
      keys[i] = key; offs[i] = pts.size();
      for (const auto& ip : tmp) { mm_interval_point q; std::memset(&q, 0, sizeof q); q.pos = ip.pos; q.hash = ip.hash; q.seqId = ip.seqId; q.side = ip.side; pts.push_back(q); }
    }
    offs[nKeys] = pts.size();
    std::vector<int32_t> clen(metadata.size());
    for (size_t i = 0; i < metadata.size(); i++) clen[i] = metadata[i].len;
    if (mm_index_upload_full(ctx_, reinterpret_cast<const mm_minmer*>(all.data()), all.size(), keys.data(), offs.data(), nKeys, pts.data(), pts.size(),
                             clen.data(), param.skip_prefix ? groups.data() : nullptr, metadata.size(), param.kmer_pct_threshold) != MM_OK)
      die("mm_index_upload_full");
  }

  void build() {      // winSketch.hpp:147-231 + :379-504, with the compute moved behind mm_index_build
    std::unordered_set<std::string> allowed;
    if (!param.target_list.empty()) {
      std::ifstream fl(param.target_list);
      std::string name;
      while (std::getline(fl, name)) allowed.insert(name);
    }
    // the reference FASTA files, parsed by param.threads workers (seq_parse.hpp) into one page-locked buffer per file; --loadIndex
    // still reads them for names and lengths (:181-211)
    std::vector<int64_t> offs(1, 0);
    seqno_t seqCounter = 0;
    std::vector<mmhost::ParsedBatch> parts;
    const bool needBases = param.loadIndexFilename.empty();
    for (const auto& fileName : param.refSequences) {
      mmhost::BatchReader rd({fileName}, (size_t)-1 >> 1, (unsigned)std::min(16, std::max(1, param.threads)), allowed, param.target_prefix,   // see skch_map.hpp: more workers than this get in each other's way
                             [](size_t n) { return (char*)mm_host_alloc(n); }, [](char* p) { mm_host_free(p); });
      mmhost::ParsedBatch b;
      while (rd.next(b)) {
        for (size_t r = 0; r < b.size(); r++) {
          // offset_t is int32 here as in the reference's default build (base_types.hpp:17-22); its -DLARGE_CONTIG variant (int64
          // coordinates, CMakeLists.txt:23) has no counterpart in the device layouts, so such a contig is refused, not truncated
          if (b.offs[r + 1] - b.offs[r] > (int64_t)std::numeric_limits<offset_t>::max()) {
            std::cerr << "[mashmap::skch::Sketch::build] ERROR: reference sequence " << b.names[r] << " has " << (b.offs[r + 1] - b.offs[r])
                      << " bp; contigs of more than " << std::numeric_limits<offset_t>::max()
                      << " bp need the reference's LARGE_CONTIG build (64-bit offset_t), which mashmap_hip does not provide" << std::endl;
            exit(1);
          }
          metadata.push_back(ContigInfo{b.names[r], (offset_t)(b.offs[r + 1] - b.offs[r])});
          seqCounter++;
        }
        parts.push_back(std::move(b));
        b = mmhost::ParsedBatch();
      }
      sequencesByFileInfo.push_back(seqCounter);
    }
    // one contiguous buffer for mm_index_build (a single part is used as it is)
    char* bases = nullptr; bool ownBases = false;
    if (needBases) {
      int64_t total = 0;
      for (auto& p : parts) total += p.totalBases();
      if (parts.size() == 1) bases = parts[0].bases;
      else { bases = (char*)mm_host_alloc((size_t)total + 64); ownBases = true; if (!bases) { std::cerr << "[mashmap_hip::skch::Sketch] ERROR: out of page-locked host memory" << std::endl; exit(1); } }
      int64_t at = 0;
      for (auto& p : parts) {
        if (ownBases && p.totalBases()) std::memcpy(bases + at, p.bases, (size_t)p.totalBases());
        for (size_t r = 0; r < p.size(); r++) offs.push_back(at + p.offs[r + 1]);
        at += p.totalBases();
      }
    } else for (size_t i = 0; i < metadata.size(); i++) offs.push_back(0);
    auto releaseParts = [&]() { for (auto& p : parts) if (p.bases) { mm_host_free(p.bases); p.bases = nullptr; } if (ownBases && bases) mm_host_free(bases); bases = nullptr; };
    if (seqCounter == 0) {
      std::cerr << "[mashmap::skch::Sketch::build] ERROR: No sequences indexed!" << std::endl;
      exit(1);
    }
    std::vector<int> groups;
    if (param.skip_prefix) groups = refGroups();
    if (!param.loadIndexFilename.empty()) { this->loadIndex(groups); releaseParts(); }
    else if (mm_index_build(ctx_, bases ? bases : "", offs.data(), metadata.size(), param.skip_prefix ? groups.data() : nullptr,
                            param.kmer_pct_threshold) != MM_OK) die("mm_index_build");
    releaseParts();
    size_t nM, nK, nP, nF; int32_t ft;
    if (mm_index_sizes(ctx_, &nM, &nK, &nP, &nF, &ft) != MM_OK) die("mm_index_sizes");
    frequentSeeds.resize(nF);
    if (mm_index_download(ctx_, nullptr, nullptr, nullptr, nullptr, frequentSeeds.data()) != MM_OK) die("mm_index_download");
    std::sort(frequentSeeds.begin(), frequentSeeds.end());
    freqThreshold = ft;
    nMinmers_ = nM;
    // minmerIndex in reference layout is 24 bytes per record (8.7 GB for a 3 Gbp reference) that nothing on the device path reads: it is
    // brought back from the device when somebody asks (materializeMinmerIndex / searchIndex), or right away with MASHMAP_HIP_EAGER_INDEX=1
    if (getenv("MASHMAP_HIP_EAGER_INDEX")) materializeMinmerIndex();
    if (getenv("MASHMAP_HIP_TIMING")) {
      mm_index_layout lay;
      if (mm_index_layout_get(ctx_, &lay) == MM_OK)
        std::cerr << "[mashmap_hip::timing] index layout: seed table " << lay.seedTableSlots << " slots (" << (lay.seedTableBytes >> 20) << " MiB), tagged=" << lay.tagged
                  << ", tag bytes " << lay.tagBytes << ", filter bytes " << lay.filterBytes << ", events " << lay.events << ", open records " << lay.openRecords << std::endl;
    }
    std::cerr << "[mashmap::skch::Sketch::build] minmer windows picked from reference (after frequent-seed removal) = " << nM << std::endl;
    std::cerr << "[mashmap::skch::Sketch::index] unique minmers = " << nK << std::endl;
    if (nF == 0) std::cerr << "[mashmap::skch::Sketch::computeFreqHist] With threshold " << param.kmer_pct_threshold
                           << "%, consider all minmers during lookup." << std::endl;
    else std::cerr << "[mashmap::skch::Sketch::computeFreqHist] With threshold " << param.kmer_pct_threshold << "%, ignore minmers with more than >= "
                   << freqThreshold << " interval points during mapping." << std::endl;
  }
};

}  // namespace skch

/*
 * oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin extern "C" shell around the *unmodified* reference headers under
 * /root/reference/src (compiled where they lie; nothing is copied into this repo).
 * Built by oracle/Makefile into oracle/_ref/libmashmap_ref.so when /root/reference is
 * present.  It exists so that
 *   (1) the CPU restatement in oracle/*.cpp can be pinned against the real reference, and
 *   (2) golden fixtures under tests/golden/ can be generated (tests/golden/make_golden.py).
 *
 * Everything callable here is a direct call into reference code:
 *   CommonFunc::getHash            src/map/include/commonFunc.hpp:138
 *   CommonFunc::sketchSequence     src/map/include/commonFunc.hpp:183
 *   CommonFunc::addMinmers         src/map/include/commonFunc.hpp:302
 *   Sketch::Sketch                 src/map/include/winSketch.hpp:122
 *   Map::doL1Mapping               src/map/include/computeMap.hpp:1130
 *   Map::computeL2MappedRegions    src/map/include/computeMap.hpp:1276
 *   Map::mapSingleQueryFrag        src/map/include/computeMap.hpp:756
 *   Map::mapModule                 src/map/include/computeMap.hpp:570
 *   Stat::*                        src/map/include/map_stats.hpp:45-262
 */
#include <bits/stdc++.h>
#include <filesystem>
#include <zlib.h>
#include <pthread.h>
#include <unistd.h>

#define private public
#include "map/include/map_parameters.hpp"
#include "map/include/base_types.hpp"
#include "map/include/winSketch.hpp"
#include "map/include/computeMap.hpp"
#undef private

extern "C" {

struct rh_minmer  { uint64_t hash; int32_t wpos, wpos_end, seqId; int16_t strand; int16_t pad; };
struct rh_point   { int32_t pos; int32_t pad0; uint64_t hash; int32_t seqId; int8_t side; int8_t pad1[3]; };
struct rh_l1      { int32_t seqId, rangeStartPos, rangeEndPos, intersectionSize; };
struct rh_l2      { int32_t seqId, meanOptimalPos, optimalStart, optimalEnd, sharedSketchSize, strand; };
struct rh_mapping {
  int32_t queryLen, refStartPos, refEndPos, queryStartPos, queryEndPos, refSeqId, querySeqId, blockLength;
  float nucIdentity, nucIdentityUpperBound;
  int32_t sketchSize, conservedSketches, strand, approxMatches;
  double kmerComplexity;
};

static_assert(sizeof(rh_minmer) == 24 && sizeof(skch::MinmerInfo) == 24, "MinmerInfo layout");

uint64_t ref_get_hash(const char* s, int len) { return skch::CommonFunc::getHash(s, len); }

static void put(rh_minmer& o, const skch::MinmerInfo& m) {
  o.hash = m.hash; o.wpos = m.wpos; o.wpos_end = m.wpos_end; o.seqId = m.seqId; o.strand = m.strand; o.pad = 0;
}

int ref_sketch_sequence(const char* seq, int len, int k, int s, int seqId, rh_minmer* out, int cap) {
  std::string buf(seq, seq + len);
  std::vector<skch::MinmerInfo> v;
  skch::CommonFunc::sketchSequence(v, &buf[0], len, k, 4, s, seqId);
  int n = (int)v.size();
  for (int i = 0; i < n && i < cap; i++) put(out[i], v[i]);
  return n;
}

int64_t ref_add_minmers(const char* seq, int len, int k, int w, int s, int seqId, rh_minmer* out, int64_t cap) {
  std::string buf(seq, seq + len);
  std::vector<skch::MinmerInfo> v;
  skch::CommonFunc::addMinmers(v, &buf[0], len, k, w, 4, s, seqId);
  int64_t n = (int64_t)v.size();
  for (int64_t i = 0; i < n && i < cap; i++) put(out[i], v[i]);
  return n;
}

float ref_j2md(float j, int k) { return skch::Stat::j2md(j, k); }
float ref_md2j(float d, int k) { return skch::Stat::md2j(d, k); }
float ref_md_lower_bound(float d, int s, int k, float ci) { return skch::Stat::md_lower_bound(d, s, k, ci); }
int ref_min_hits(int s, int k, float pi) { return skch::Stat::estimateMinimumHits(s, k, pi); }
int ref_min_hits_relaxed(int s, int k, float pi) {
  return skch::Stat::estimateMinimumHitsRelaxed(s, k, pi, skch::fixed::confidence_interval);
}
int64_t ref_recommended_sketch_size(int k, float pi, int64_t segLength, uint64_t refSize) {
  return skch::Stat::recommendedSketchSize(skch::fixed::pval_cutoff, skch::fixed::confidence_interval,
                                           k, 4, pi, segLength, refSize);
}

/* ---- session: a reference Sketch + Map built exactly as mash_map.cpp:43,51 does ---- */
struct RefSession {
  skch::Parameters p;
  skch::Sketch* sketch = nullptr;
  skch::Map* map = nullptr;
  std::string dummyQuery, dummyOut;
};

enum { RF_HG = 1, RF_SKIP_SELF = 2, RF_SKIP_PREFIX = 4, RF_LOWER_TRI = 8, RF_NOSPLIT = 16, RF_NOMERGE = 32,
       RF_DROP_LOW_ID = 64 };

void* ref_session_new(const char* refFilesNl, int k, int segLength, int sketchSize, float pi,
                      int filterMode, int flags, char prefixDelim, float kmerPctThreshold,
                      int numMappings, int threads) {
  auto* S = new RefSession();
  skch::Parameters& p = S->p;
  std::stringstream ss(refFilesNl); std::string f;
  while (std::getline(ss, f, '\n')) if (!f.empty()) p.refSequences.push_back(f);
  char tmpl[] = "/tmp/rh_dummyXXXXXX"; int fd = mkstemp(tmpl); if (fd >= 0) { (void)!write(fd, ">d\nACGT\n", 8); close(fd); }
  S->dummyQuery = tmpl; S->dummyOut = std::string(tmpl) + ".out";
  p.querySequences.push_back(S->dummyQuery);
  p.outFileName = S->dummyOut;
  /* same defaults parseandSave() installs (parseCmdArgs.hpp:257-655) */
  p.kmerSize = k; p.segLength = segLength; p.sketchSize = sketchSize; p.percentageIdentity = pi;
  p.alphabetSize = 4; p.referenceSize = 0;
  p.block_length = segLength; p.chain_gap = segLength;
  p.kmer_pct_threshold = kmerPctThreshold;
  p.stage2_full_scan = true;
  p.stage1_topANI_filter = (flags & RF_HG) != 0;
  p.ANIDiff = skch::fixed::ANIDiff; p.ANIDiffConf = skch::fixed::ANIDiffConf;
  p.filterMode = filterMode;
  p.numMappingsForSegment = numMappings; p.numMappingsForShortSequence = numMappings;
  p.threads = threads;
  p.split = !(flags & RF_NOSPLIT);
  p.lower_triangular = (flags & RF_LOWER_TRI) != 0;
  p.skip_self = (flags & RF_SKIP_SELF) != 0;
  p.skip_prefix = (flags & RF_SKIP_PREFIX) != 0;
  p.prefix_delim = (flags & RF_SKIP_PREFIX) ? prefixDelim : '\0';
  p.mergeMappings = !(flags & RF_NOMERGE);
  p.keep_low_pct_id = !(flags & RF_DROP_LOW_ID);
  p.report_ANI_percentage = false; p.filterLengthMismatches = false;
  p.kmerComplexityThreshold = 0.0;
  p.use_spaced_seeds = false; p.world_minimizers = false; p.spaced_seed_sensitivity = 0;
  p.sparsity_hash_threshold = std::numeric_limits<uint64_t>::max();
  p.legacy_output = false;
  S->sketch = new skch::Sketch(p);
  S->map = new skch::Map(p, *S->sketch);
  return S;
}

void ref_session_free(void* h) {
  auto* S = (RefSession*)h;
  unlink(S->dummyQuery.c_str()); unlink(S->dummyOut.c_str());
  delete S->map; delete S->sketch; delete S;
}

int64_t ref_session_index_size(void* h) { return (int64_t)((RefSession*)h)->sketch->minmerIndex.size(); }
void ref_session_index_copy(void* h, rh_minmer* out) {
  auto& v = ((RefSession*)h)->sketch->minmerIndex;
  for (size_t i = 0; i < v.size(); i++) put(out[i], v[i]);
}
int64_t ref_session_nkeys(void* h) { return (int64_t)((RefSession*)h)->sketch->minmerPosLookupIndex.size(); }
/* dumps keys in the map's iteration order together with their point counts */
void ref_session_keys(void* h, uint64_t* keys, int64_t* counts) {
  size_t i = 0;
  for (auto& e : ((RefSession*)h)->sketch->minmerPosLookupIndex) { keys[i] = e.first; counts[i] = (int64_t)e.second.size(); i++; }
}
int64_t ref_session_lookup(void* h, uint64_t hash, rh_point* out, int64_t cap) {
  auto& m = ((RefSession*)h)->sketch->minmerPosLookupIndex;
  auto it = m.find(hash);
  if (it == m.end()) return -1;
  int64_t n = (int64_t)it->second.size();
  for (int64_t i = 0; i < n && i < cap; i++) {
    const auto& ip = it->second[i];
    std::memset(&out[i], 0, sizeof(rh_point));
    out[i].pos = ip.pos; out[i].hash = ip.hash; out[i].seqId = ip.seqId; out[i].side = ip.side;
  }
  return n;
}
int ref_session_is_freq(void* h, uint64_t hash) { return ((RefSession*)h)->sketch->isFreqSeed(hash) ? 1 : 0; }
int ref_session_freq_threshold(void* h) { return ((RefSession*)h)->sketch->getFreqThreshold(); }
int ref_session_ncontigs(void* h) { return (int)((RefSession*)h)->sketch->metadata.size(); }
int ref_session_contig_len(void* h, int i) { return ((RefSession*)h)->sketch->metadata[i].len; }
int ref_session_ncutoffs(void* h) { return (int)((RefSession*)h)->map->sketchCutoffs.size(); }
void ref_session_cutoffs(void* h, int* out) {
  auto& v = ((RefSession*)h)->map->sketchCutoffs;
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
}

/*
 * One query fragment through the reference's L1 and L2 (computeMap.hpp:756-815).
 * Outputs (each with a capacity and a returned count in counts[]):
 *   counts[0] sketch (after frequent-seed removal, computeMap.hpp:834-839) -> qsk
 *   counts[1] interval points (computeMap.hpp:857)                         -> pts
 *   counts[2] L1 candidates in computeL1CandidateRegions order             -> l1
 *   counts[3] L2 loci for *every* L1 candidate, candidate-major            -> l2 (l2cand[i] = candidate index)
 *   counts[4] final l2Mappings of mapSingleQueryFrag                       -> maps
 *   counts[5] minimumHits (estimateMinimumHitsRelaxed for Q.sketchSize)
 *   counts[6] Q.sketchSize ; counts[7] raw sketch size before frequent-seed removal
 */
int ref_session_map_fragment(void* h, const char* seq, int len, int fullLen, int seqCounter, const char* seqName,
                             rh_minmer* qsk, int qskCap, rh_point* pts, int ptsCap, rh_l1* l1, int l1Cap,
                             rh_l2* l2, int* l2cand, int l2Cap, rh_mapping* maps, int mapsCap,
                             int64_t* counts, double* kmerComplexity) {
  auto* S = (RefSession*)h;
  skch::Map& M = *S->map;
  typedef skch::Sketch::MI_Type MinVec;
  std::string buf(seq, seq + len);
  {
    skch::QueryMetaData<MinVec> Q;
    Q.seq = &buf[0]; Q.len = len; Q.fullLen = fullLen; Q.seqCounter = seqCounter; Q.seqName = seqName;
    Q.refGroup = M.getRefGroup(Q.seqName);
    Q.sketchSize = 0; Q.kmerComplexity = 0;
    std::vector<skch::IntervalPoint> ip;
    std::vector<skch::Map::L1_candidateLocus_t> l1v;
    M.doL1Mapping(Q, ip, l1v);
    counts[0] = (int64_t)Q.minmerTableQuery.size();
    for (int i = 0; i < (int)Q.minmerTableQuery.size() && i < qskCap; i++) put(qsk[i], Q.minmerTableQuery[i]);
    counts[1] = (int64_t)ip.size();
    for (int i = 0; i < (int)ip.size() && i < ptsCap; i++) {
      std::memset(&pts[i], 0, sizeof(rh_point));
      pts[i].pos = ip[i].pos; pts[i].hash = ip[i].hash; pts[i].seqId = ip[i].seqId; pts[i].side = ip[i].side;
    }
    counts[2] = (int64_t)l1v.size();
    for (int i = 0; i < (int)l1v.size() && i < l1Cap; i++)
      l1[i] = rh_l1{l1v[i].seqId, l1v[i].rangeStartPos, l1v[i].rangeEndPos, l1v[i].intersectionSize};
    int64_t nl2 = 0;
    for (int c = 0; c < (int)l1v.size(); c++) {
      std::vector<skch::Map::L2_mapLocus_t> loci;
      auto cand = l1v[c];
      M.computeL2MappedRegions(Q, cand, loci);
      for (auto& x : loci) {
        if (nl2 < l2Cap) { l2[nl2] = rh_l2{x.seqId, x.meanOptimalPos, x.optimalStart, x.optimalEnd, x.sharedSketchSize, x.strand}; l2cand[nl2] = c; }
        nl2++;
      }
    }
    counts[3] = nl2;
    counts[5] = Q.sketchSize > 0 ? skch::Stat::estimateMinimumHitsRelaxed(Q.sketchSize, S->p.kmerSize, S->p.percentageIdentity, skch::fixed::confidence_interval) : 0;
    counts[6] = Q.sketchSize;
    *kmerComplexity = Q.kmerComplexity;
  }
  {
    std::string buf2(seq, seq + len);
    std::vector<skch::MinmerInfo> raw;
    skch::CommonFunc::sketchSequence(raw, &buf2[0], len, S->p.kmerSize, 4, S->p.sketchSize, seqCounter);
    counts[7] = (int64_t)raw.size();
  }
  {
    std::string buf3(seq, seq + len);
    skch::QueryMetaData<MinVec> Q;
    Q.seq = &buf3[0]; Q.len = len; Q.fullLen = fullLen; Q.seqCounter = seqCounter; Q.seqName = seqName;
    Q.refGroup = M.getRefGroup(Q.seqName);
    Q.sketchSize = 0; Q.kmerComplexity = 0;
    std::vector<skch::IntervalPoint> ip;
    std::vector<skch::Map::L1_candidateLocus_t> l1v;
    skch::MappingResultsVector_t out;
    M.mapSingleQueryFrag(Q, ip, l1v, out);
    counts[4] = (int64_t)out.size();
    for (int i = 0; i < (int)out.size() && i < mapsCap; i++) {
      const auto& e = out[i];
      maps[i] = rh_mapping{e.queryLen, e.refStartPos, e.refEndPos, e.queryStartPos, e.queryEndPos, e.refSeqId,
                           e.querySeqId, e.blockLength, e.nucIdentity, e.nucIdentityUpperBound, e.sketchSize,
                           e.conservedSketches, (int32_t)e.strand, e.approxMatches, (double)e.kmerComplexity};
    }
  }
  return 0;
}

/* A whole read through mapModule (computeMap.hpp:570): split, L1/L2, chain-merge, filter. */
int ref_session_map_read(void* h, const char* seq, int len, int seqCounter, const char* seqName,
                         rh_mapping* maps, int mapsCap) {
  auto* S = (RefSession*)h;
  progress_meter::ProgressMeter pm(0, ""); // total 0: logger thread exits at once
  auto* in = new skch::InputSeqProgContainer(std::string(seq, seq + len), seqName, seqCounter, pm);
  skch::MapModuleOutput* out = S->map->mapModule(in);
  int n = (int)out->readMappings.size();
  for (int i = 0; i < n && i < mapsCap; i++) {
    const auto& e = out->readMappings[i];
    maps[i] = rh_mapping{e.queryLen, e.refStartPos, e.refEndPos, e.queryStartPos, e.queryEndPos, e.refSeqId,
                         e.querySeqId, e.blockLength, e.nucIdentity, e.nucIdentityUpperBound, e.sketchSize,
                         e.conservedSketches, (int32_t)e.strand, e.approxMatches, (double)e.kmerComplexity};
  }
  delete out; delete in;
  pm.finish();
  return n;
}

}  // extern "C"

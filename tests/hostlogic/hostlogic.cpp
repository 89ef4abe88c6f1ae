// tests/hostlogic/hostlogic.cpp -- TEST HARNESS: a C entry point around skch::MapPost (mashmap_amd/host/skch_map_post.hpp), the
// device-independent half of the host side, so that it can be driven on a machine without a GPU: the integers normally produced
// by the kernels are supplied by the caller (the tests take them from the real reference's own L1/L2 stages).
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../../mashmap_amd/host/skch_map_post.hpp"

extern "C" {

struct hl_mapping {      // same fields as oracle.h's orc_mapping
  int32_t queryLen, refStartPos, refEndPos, queryStartPos, queryEndPos, refSeqId, querySeqId, blockLength;
  float nucIdentity, nucIdentityUpperBound;
  int32_t sketchSize, conservedSketches, strand, approxMatches;
  double kmerComplexity;
};

enum { HL_HG = 1, HL_SKIP_SELF = 2, HL_SKIP_PREFIX = 4, HL_LOWER_TRI = 8, HL_NOSPLIT = 16, HL_NOMERGE = 32, HL_DROP_LOW_ID = 64 };

int hl_map_read(int k, int segLength, int sketchSize, float pi, int filterMode, int flags, int numMappings,
                int nContigs, const char* const* names, const int32_t* lens, const int32_t* groups,
                const char* readName, int readLen, int seqCounter,
                int nFrags, const mm_fragment* frags, const mm_frag_stats* stats, int nL1, const mm_l1_candidate* l1, int nL2, const mm_l2_locus* l2,
                hl_mapping* out, int cap, char* paf, int pafCap) {
  skch::Parameters p;
  p.kmerSize = k; p.segLength = segLength; p.block_length = segLength; p.chain_gap = segLength; p.sketchSize = sketchSize;
  p.percentageIdentity = pi; p.filterMode = filterMode; p.stage1_topANI_filter = (flags & HL_HG) != 0;
  p.skip_self = (flags & HL_SKIP_SELF) != 0; p.skip_prefix = (flags & HL_SKIP_PREFIX) != 0; p.lower_triangular = (flags & HL_LOWER_TRI) != 0;
  p.split = !(flags & HL_NOSPLIT); p.mergeMappings = !(flags & HL_NOMERGE); p.keep_low_pct_id = !(flags & HL_DROP_LOW_ID);
  p.numMappingsForSegment = (uint32_t)numMappings; p.numMappingsForShortSequence = (uint32_t)numMappings;
  std::vector<skch::ContigInfo> meta;
  for (int i = 0; i < nContigs; i++) meta.push_back(skch::ContigInfo{names[i], lens[i]});
  std::vector<int> grp;
  if (groups) grp.assign(groups, groups + nContigs);
  skch::MapPost post(p, meta, grp);
  skch::DeviceResults D;
  D.frags.assign(frags, frags + nFrags); D.stats.assign(stats, stats + nFrags);
  D.l1.assign(l1, l1 + nL1); D.l2.assign(l2, l2 + nL2);
  D.fragBegin = {0, (size_t)nFrags};
  D.l1Begin.resize((size_t)nFrags + 1);
  { size_t o = 0; for (int f = 0; f < nFrags; f++) { D.l1Begin[f] = o; o += (size_t)stats[f].nL1; } D.l1Begin[nFrags] = o; }
  D.l2Begin.assign((size_t)nL1 + 1, (size_t)nL2);
  { size_t i = 0; for (int c = 0; c <= nL1; c++) { while (i < (size_t)nL2 && D.l2[i].cand < c) i++; D.l2Begin[c] = i; } }
  skch::MappingResultsVector_t res;
  post.mapModule(D, 0, readName, readLen, seqCounter, res);
  std::ostringstream os;
  post.reportReadMappings(res, readName, os);
  const std::string txt = os.str();
  if (paf && pafCap > 0) { std::strncpy(paf, txt.c_str(), (size_t)pafCap - 1); paf[pafCap - 1] = 0; }
  for (size_t i = 0; i < res.size() && (int)i < cap; i++) {
    const auto& e = res[i];
    out[i] = hl_mapping{e.queryLen, e.refStartPos, e.refEndPos, e.queryStartPos, e.queryEndPos, e.refSeqId, e.querySeqId, e.blockLength,
                        e.nucIdentity, e.nucIdentityUpperBound, e.sketchSize, e.conservedSketches, (int32_t)e.strand, e.approxMatches,
                        (double)e.kmerComplexity};
  }
  return (int)res.size();
}

}  // extern "C"

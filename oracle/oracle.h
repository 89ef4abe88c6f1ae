/*
 * oracle/oracle.h -- C ABI of the CPU restatement of MashMap's sketch + L1/L2 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle.so, and only as the checker / reported baseline.
 * The product (mashmap_amd/, include/) never includes, links or calls anything here.
 *
 * Parity status: PINNED against the real reference compiled from /root/reference
 * (oracle/_ref, see oracle/Makefile + tests/test_oracle_vs_ref.py) and against the golden
 * vectors in tests/golden/ that were generated from it (tests/golden/make_golden.py).
 * The only un-pinned boundary is GNU GSL (absent; see gsl_shim/gsl/gsl_cdf.h).
 */
#ifndef MASHMAP_ORACLE_H
#define MASHMAP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same 24-byte layout as skch::MinmerInfo (base_types.hpp:31) */
typedef struct { uint64_t hash; int32_t wpos, wpos_end, seqId; int16_t strand; int16_t pad; } orc_minmer;
/* same field order as skch::IntervalPoint (base_types.hpp:66) */
typedef struct { int32_t pos; int32_t pad0; uint64_t hash; int32_t seqId; int8_t side; int8_t pad1[3]; } orc_point;
/* Map::L1_candidateLocus_t (computeMap.hpp:58) */
typedef struct { int32_t seqId, rangeStartPos, rangeEndPos, intersectionSize; } orc_l1;
/* Map::L2_mapLocus_t (computeMap.hpp:76) */
typedef struct { int32_t seqId, meanOptimalPos, optimalStart, optimalEnd, sharedSketchSize, strand; } orc_l2;
/* the fields of skch::MappingResult (base_types.hpp:154) that reach the PAF line */
typedef struct {
  int32_t queryLen, refStartPos, refEndPos, queryStartPos, queryEndPos, refSeqId, querySeqId, blockLength;
  float nucIdentity, nucIdentityUpperBound;
  int32_t sketchSize, conservedSketches, strand, approxMatches;
  double kmerComplexity;
} orc_mapping;

enum { ORC_HG = 1, ORC_SKIP_SELF = 2, ORC_SKIP_PREFIX = 4, ORC_LOWER_TRI = 8, ORC_NOSPLIT = 16, ORC_NOMERGE = 32,
       ORC_DROP_LOW_ID = 64 };

/* a3: MurmurHash3_x64_128 low word, seed 42 (commonFunc.hpp:138, murmur3.h:226) */
uint64_t orc_get_hash(const char* s, int len);
/* a1: in-place normalisation (commonFunc.hpp:97) */
void orc_normalise(char* seq, int64_t len);
/* a4: query fragment sketch (commonFunc.hpp:183) */
int orc_sketch_sequence(const char* seq, int len, int k, int s, int seqId, orc_minmer* out, int cap);
/* a5: reference minmer intervals of one contig (commonFunc.hpp:302) */
int64_t orc_add_minmers(const char* seq, int len, int k, int w, int s, int seqId, orc_minmer* out, int64_t cap);

/* a14: float statistics (map_stats.hpp:45-262) */
float orc_j2md(float j, int k);
float orc_md2j(float d, int k);
float orc_md_lower_bound(float d, int s, int k, float ci);
int orc_min_hits(int s, int k, float pi);
int orc_min_hits_relaxed(int s, int k, float pi);
int64_t orc_recommended_sketch_size(int k, float pi, int64_t segLength, uint64_t refSize);

/* a session = reference index (a5-a7) + mapper parameters, built from in-memory contigs */
void* orc_session_new(int k, int segLength, int sketchSize, float pi, int filterMode, int flags,
                      char prefixDelim, float kmerPctThreshold, int numMappings);
/* contigs must be added in file order; name is the FASTA header up to the first space */
void orc_session_add_contig(void* h, const char* name, const char* seq, int len);
/* replaces minmerIndex (sorted by (seqId, wpos)) before orc_session_finalize */
void orc_session_set_index(void* h, const orc_minmer* recs, int64_t n);
/* runs Sketch::index + frequency filter (winSketch.hpp:379-504) and Map::setProbs (computeMap.hpp:178) */
void orc_session_finalize(void* h);
void orc_session_free(void* h);

int64_t orc_session_index_size(void* h);
void orc_session_index_copy(void* h, orc_minmer* out);
int64_t orc_session_nkeys(void* h);
void orc_session_keys(void* h, uint64_t* keys, int64_t* counts); /* ascending key order */
int64_t orc_session_lookup(void* h, uint64_t hash, orc_point* out, int64_t cap);
int64_t orc_session_npoints(void* h);
void orc_session_export_lookup(void* h, uint64_t* keys, uint64_t* offsets, orc_point* pts);
int64_t orc_session_nfreq(void* h);
void orc_session_freq_list(void* h, uint64_t* out);
int orc_session_is_freq(void* h, uint64_t hash);
int orc_session_freq_threshold(void* h);
int orc_session_ncontigs(void* h);
int orc_session_contig_len(void* h, int i);
int orc_session_ncutoffs(void* h);
void orc_session_cutoffs(void* h, int* out);

/* same contract as ref_session_map_fragment in ref_harness.cpp */
int orc_session_map_fragment(void* h, const char* seq, int len, int fullLen, int seqCounter, const char* seqName,
                             orc_minmer* qsk, int qskCap, orc_point* pts, int ptsCap, orc_l1* l1, int l1Cap,
                             orc_l2* l2, int* l2cand, int l2Cap, orc_mapping* maps, int mapsCap,
                             int64_t* counts, double* kmerComplexity);
/* the literal L1 (computeMap.hpp:916-1116) on a caller-made, sorted point list */
int orc_session_l1_from_points(void* h, const orc_point* pts, int64_t n, int qSketchSize, int fragLen, int minimumHits, orc_l1* out, int cap);
/* a whole read through the mapModule logic (computeMap.hpp:570-714) */
int orc_session_map_read(void* h, const char* seq, int len, int seqCounter, const char* seqName,
                         orc_mapping* maps, int mapsCap);

#ifdef __cplusplus
}
#endif
#endif

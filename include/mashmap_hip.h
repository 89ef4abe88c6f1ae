/*
 * include/mashmap_hip.h -- C ABI of libmashmap_hip.so: MashMap's sketch + L1/L2 hot path as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (marbl/MashMap v3.1.3) has no FFI: its boundary is the two C++ classes main()
 * constructs (src/map/mash_map.cpp:43,51) and the free functions they call.  Every entry point
 * below names the reference interface it stands in for (file:line relative to the reference
 * tree).  Plain pointers and sizes only; the library owns all device memory; the caller owns
 * every host buffer.  All functions return 0 on success or a negative MM_ERR_* code
 * (mm_last_error() gives the text); nothing here ever falls back to a CPU implementation --
 * if no gfx950 device / kernel image is available mm_create() fails.
 *
 * Thread model: one mm_ctx per GPU (one process per GPU in multi-GPU runs); a ctx is
 * thread-compatible, not thread-safe; all work of a ctx is ordered on its own HIP stream.
 */
#ifndef MASHMAP_HIP_H
#define MASHMAP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 2     /* 2: mm_pass_stats writes 5 counts; mm_reads_prefetch_packed_append reports whether it staged; mm_reads_prefetch_drop */

enum {
  MM_OK = 0,
  MM_ERR_ARG = -1,        /* bad argument / unsupported parameter combination */
  MM_ERR_DEVICE = -2,     /* HIP runtime error (no device, launch failure, out of memory ...) */
  MM_ERR_STATE = -3,      /* call sequence violated (e.g. map before index upload) */
  MM_ERR_CAPACITY = -4    /* an internal device buffer overflowed and could not be grown */
};

/* skch::Parameters fields the hot path reads (map_parameters.hpp:32-80) */
enum {
  MM_FLAG_HG_FILTER = 1,        /* stage1_topANI_filter   (parseCmdArgs.hpp:397) */
  MM_FLAG_SKIP_SELF = 2,        /* skip_self              (parseCmdArgs.hpp:341) */
  MM_FLAG_SKIP_PREFIX = 4,      /* skip_prefix            (parseCmdArgs.hpp:349) */
  MM_FLAG_LOWER_TRIANGULAR = 8, /* lower_triangular       (parseCmdArgs.hpp:334) */
  MM_FLAG_NO_SPLIT = 16         /* !split                 (parseCmdArgs.hpp:427): a read longer than segLength is then ONE fragment with
                                   windowLen = len - segLength != 0 (computeMap.hpp:933, :1309); a batch that holds such a read goes through
                                   the literal kernels (k_l1_window, k_l2_window: exact, not fast).  A read of any length: one that does not
                                   fit a CU's LDS (a whole contig as a read) is sketched by the exact kernel from global memory. */
};

typedef struct {
  int32_t kmerSize;       /* Parameters::kmerSize   (1..64; 16..32 take the tuned strip hasher, 19 -- the reference's default -- the tuned tail as well) */
  int32_t segLength;      /* Parameters::segLength  */
  int32_t sketchSize;     /* Parameters::sketchSize (1 .. 65535.  Up to 8190 the sketch and L2 kernels keep their state in a CU's 160 KB of LDS -- mm_create
                             checks the combination with segLength and says what does not fit --; beyond, every fragment takes a global-memory sketch
                             kernel and the literal L2 kernels, and the index build keeps a window's sketch in HBM from 4097 on: exact, not tuned) */
  int32_t flags;          /* MM_FLAG_* */
} mm_params;

/* bit-for-bit skch::MinmerInfo (base_types.hpp:31; 24 bytes) */
typedef struct { uint64_t hash; int32_t wpos, wpos_end, seqId; int16_t strand; int16_t pad_; } mm_minmer;
/* bit-for-bit skch::IntervalPoint (base_types.hpp:66; 24 bytes) */
typedef struct { int32_t pos; int32_t pad0_; uint64_t hash; int32_t seqId; int8_t side; int8_t pad1_[3]; } mm_interval_point;

/* one query fragment, as Map::mapModule cuts them (computeMap.hpp:587-671) */
typedef struct {
  int32_t readId;         /* index of the read in the uploaded batch (InputSeqContainer::seqCounter - first) */
  int32_t fragStart;      /* offset of the fragment inside the read == MappingResult::queryStartPos */
  int32_t len;            /* QueryMetaData::len */
  int32_t pad_;
} mm_fragment;

/* per-fragment integers behind QueryMetaData (base_types.hpp:265) after getSeedHits (computeMap.hpp:818) */
typedef struct {
  int32_t rawSketchSize;  /* |minmerTableQuery| before frequent-seed removal (feeds kmerComplexity, :830) */
  int32_t sketchSize;     /* Q.sketchSize (:839) */
  uint64_t maxHash;       /* minmerTableQuery.back().hash before removal (:830) */
  int32_t nPoints;        /* interval points that survived the seqId filters (:891-896) */
  int32_t nL1;            /* L1 candidates (:916) */
} mm_frag_stats;

/* Map::L1_candidateLocus_t (computeMap.hpp:58) + owning fragment */
typedef struct { int32_t frag; int32_t seqId, rangeStartPos, rangeEndPos, intersectionSize; } mm_l1_candidate;
/* Map::L2_mapLocus_t (computeMap.hpp:76) + owning fragment / candidate.  `cand` as returned by mm_results_download is the index
 * into the L1 array downloaded with it.  In the DEVICE-resident records (mm_results_device / mm_results_copy_device) `cand` is the
 * candidate's rank among the candidates of its own fragment (0 .. stats[frag].nL1-1), and records are grouped by candidate but
 * not sorted by fragment. */
typedef struct {
  int32_t frag, cand;
  int32_t seqId, meanOptimalPos, optimalStart, optimalEnd, sharedSketchSize, strand;
} mm_l2_locus;

/*
 * A candidate mapping: one L2 locus that doL2Mapping reports for a fragment (computeMap.hpp:1221-1250), in integers.  The floats of
 * MappingResult (base_types.hpp:154) are functions of these: nucIdentity / nucIdentityUpperBound of (conservedSketches, sketchSize, k)
 * (map_stats.hpp:45,81), kmerComplexity of (rawSketchSize, maxHash, fragLen, k) (computeMap.hpp:830-831).  This is the record the
 * host side chains and filters, and the record multi-GPU runs exchange (mm_allgatherv_mappings).  48 bytes.
 */
typedef struct {
  int32_t querySeqId;          /* MappingResult::querySeqId == seqCounterBase + index of the read in the batch */
  int32_t fragStart, fragLen;  /* offset of the fragment in the read (queryStartPos after :632-636), Q.len */
  int32_t refSeqId, refStartPos;          /* L2_mapLocus_t::seqId, meanOptimalPos */
  int32_t conservedSketches, sketchSize;  /* sharedSketchSize, Q.sketchSize */
  int32_t strand;
  int32_t rawSketchSize, pad_;
  uint64_t maxHash;
} mm_mapping;

typedef struct mm_ctx mm_ctx;

/* ------------------------------------------------------------------------------------------ */
int  mm_abi_version(void);
/* creates the context on HIP device `device`; fails (MM_ERR_DEVICE) when there is none */
int  mm_create(mm_ctx** out, int device, const mm_params* params);
void mm_destroy(mm_ctx* ctx);
const char* mm_last_error(const mm_ctx* ctx);   /* ctx may be NULL for mm_create failures */

/*
 * Reference index -> device.  Stands in for how Map reads `const skch::Sketch&`:
 *   minmerIndex            (winSketch.hpp:102; read by computeL2MappedRegions, computeMap.hpp:1284-1340)
 *   minmerPosLookupIndex   (winSketch.hpp:101; read by getSeedIntervalPoints, computeMap.hpp:878)
 *   isFreqSeed()           (winSketch.hpp:506; read by getSeedHits, computeMap.hpp:835)
 *   metadata[i].len        (winSketch.hpp:79)
 * `minmers` is minmerIndex *after* dropFreqSeedSet (:497).  The lookup map is passed flattened:
 * keys[i] owns points[offsets[i] .. offsets[i+1]) in the map's own per-key order.
 * refGroup[i] is Map::refIdGroup (computeMap.hpp:113), or NULL when skip_prefix is off.
 */
int mm_index_upload(mm_ctx* ctx,
                    const mm_minmer* minmers, size_t nMinmers,
                    const uint64_t* keys, const uint64_t* offsets, size_t nKeys,
                    const mm_interval_point* points, size_t nPoints,
                    const uint64_t* freqSeeds, size_t nFreq,
                    const int32_t* contigLen, const int32_t* refGroup, size_t nContigs);

/*
 * Host-computed integer tables (they depend on GSL-class floating point, kept on the host):
 *   minHits[q]      = Stat::estimateMinimumHitsRelaxed(q, k, pi, 0.95), q = 0..sketchSize   (computeMap.hpp:1144)
 *   sketchCutoffs[] = Map::sketchCutoffs (computeMap.hpp:109,178-258), nCutoffs entries
 */
int mm_set_tables(mm_ctx* ctx, const int32_t* minHits, size_t nMinHits, const int32_t* sketchCutoffs, size_t nCutoffs);

/*
 * Integer tables for the best-first walk of doL2Mapping (computeMap.hpp:1182-1267) on the device, stride = sketchSize + 1:
 *   accept[Qs*stride + shared] = 1 iff a locus sharing `shared` of Qs sketch elements is reported  (:1221: identity or, with
 *                                keep_low_pct_id, its upper bound reaches percentageIdentity)
 *   minIsz[Qs*stride + best]   = smallest L1 intersectionSize that passes the ANI cut-off (:1192-1202) when the best reported
 *                                locus so far shares `best` elements
 * (mashmap_amd/host/mm_stats.hpp: mmhost::replayTables fills them with the reference's own float expressions.)  Without them
 * mm_map_fragments stops at the L2 loci and the mm_mappings_* calls fail with MM_ERR_STATE.
 */
int mm_set_replay_tables(mm_ctx* ctx, const uint8_t* accept, const int16_t* minIsz, size_t stride);

/* all three tables computed inside the library (mashmap_amd/host/mm_stats.hpp mirrors skch::Stat and Map::setProbs) with the
 * reference's defaults (ANIDiff 0, ANIDiffConf 0.999, keep_low_pct_id on) and uploaded */
int mm_set_tables_default(mm_ctx* ctx, float percentageIdentity);

/*
 * Host-side statistics (no device needed) -- mirrors of skch::Stat (map_stats.hpp:45-262), exported so
 * that parity tests and the C++ host side share one implementation:
 *   j2md :45, md2j :63, md_lower_bound :81, estimateMinimumHitsRelaxed :144 (ci = 0.95),
 *   recommendedSketchSize :234 (p-value 1e-3, ci 0.95, alphabet 4), Map::sketchCutoffs (computeMap.hpp:178-258)
 */
float   mm_stat_j2md(float j, int k);
float   mm_stat_md2j(float d, int k);
float   mm_stat_md_lower_bound(float d, int s, int k, float ci);
int     mm_stat_min_hits_relaxed(int s, int k, float percentageIdentity);
int64_t mm_stat_recommended_sketch_size(int k, float percentageIdentity, int64_t segLength, uint64_t referenceSize);
int     mm_stat_sketch_cutoffs(int sketchSize, int k, int hgFilter, int32_t* out, size_t cap);   /* returns entries written */
/* the tables of mm_set_replay_tables, (sketchSize+1)^2 entries each, from the reference's own float expressions (computeMap.hpp:1192-1222) */
int     mm_stat_replay_tables(int sketchSize, int k, float percentageIdentity, float ANIDiff, int keepLowPctId, uint8_t* accept, int16_t* minIsz);

/*
 * A batch of query reads -> device.  Replaces the `char* seq` handed to sketchSequence
 * (commonFunc.hpp:183): ASCII in, normalised (makeUpperCaseAndValidDNA, :97) and packed to
 * 2 bit/base + N mask on the device.  readOffsets has nReads+1 entries into `bases`.
 * Fragments are cut exactly as mapModule does (computeMap.hpp:587-671).
 * readRefGroup[r] = Map::getRefGroup(name) (computeMap.hpp:164) or NULL;
 * readSelfSeqId[r] = reference seqId whose name equals the read's name, -1 if none, or NULL
 * (this is what the skip_self test `Q.seqName != ref.name` needs, computeMap.hpp:891);
 * seqCounterBase = seqCounter of read 0 (lower_triangular compares seqCounter with seqId, :893).
 */
int mm_reads_upload(mm_ctx* ctx, const char* bases, const int64_t* readOffsets, size_t nReads,
                    const int32_t* readRefGroup, const int32_t* readSelfSeqId, int32_t seqCounterBase);
/* Optional: start the host-to-device copy of the ASCII bytes the NEXT mm_reads_upload will name -- [bases, bases + nBytes) must be exactly
 * the range that call reads, i.e. its `bases + readOffsets[0]` and `readOffsets[nReads] - readOffsets[0]` -- on a stream of the context's
 * own, so that it runs under the kernels of the batch being mapped (call it between mm_reads_upload and mm_map_fragments of the current
 * batch).  mm_reads_upload recognises the range and skips its own copy; any other upload simply discards the prefetch.  The bytes must
 * stay unchanged until that upload returns, and should be page-locked (mm_host_alloc): a copy from pageable memory does not overlap.
 * The two prefetch calls are the one exception to "a ctx is thread-compatible": they may come from another thread (a reader) while the
 * context's own thread is inside any other call; against mm_reads_upload* of the same context they are serialised inside the library. */
int mm_reads_prefetch(mm_ctx* ctx, const char* bases, size_t nBytes);
/* page-locked host memory for the `bases` of mm_reads_upload / mm_index_build: the copy to the GPU is then a single DMA at PCIe rate
 * (pageable memory is staged through a bounce buffer at a fraction of it).  Optional: any host pointer works. */
void* mm_host_alloc(size_t bytes);
void  mm_host_free(void* p);
/* same, from already device-resident ASCII (hipMalloc'ed by the caller, e.g. a torch tensor).  The bytes are read on the context's own
 * stream (mm_stream): work of the caller's stream that produces them must have completed before the call. */
int mm_reads_upload_device(mm_ctx* ctx, const void* dBases, size_t nBases, const int64_t* readOffsets, size_t nReads,
                           const int32_t* readRefGroup, const int32_t* readSelfSeqId, int32_t seqCounterBase);
/*
 * The same batch for a caller that has normalised and packed the reads itself (the layout k_pack2bit produces on the device, DESIGN.md
 * section 2): PCIe then carries 0.375 bytes per base instead of 1.  Read r has readLengths[r] bases and starts at packed base
 * P(r) = sum over earlier reads of ceil(len / 32) * 32 -- or, with readStarts != NULL, at P(r) = readStarts[r] - readStarts[0]: ascending
 * multiples of 32 with P(r+1) >= P(r) + ceil(len / 32) * 32, i.e. gaps between reads are allowed (a parser whose threads pack their
 * pieces of a file independently leaves one between pieces; the words of a gap travel with the rest and are never read as bases).
 * P(nReads) is the end of the last read:
 *   bases2  2 bit/base, A0 C1 G2 T3 (the code of an N is 0), sixteen bases per uint32, first base in the low bits; read r owns the words
 *           [P(r) / 16, P(r+1) / 16), bases behind its end are 0
 *   nmask   1 bit/base, set where makeUpperCaseAndValidDNA (commonFunc.hpp:97) leaves an 'N' (anything but A C G T a c g t); read r owns
 *           the words [P(r) / 32, P(r+1) / 32)
 *   readHasN[r] != 0 iff any mask bit of read r is set; NULL: derived from nmask here
 * mm_pack_read produces the words of one read on the host (AVX2 + BMI2 when the CPU has them; mm_pack_read_portable is the plain loop,
 * same result): 2 * ceil(len / 32) code words and ceil(len / 32) mask words; returns the number of N bases.
 * mm_reads_prefetch_packed is mm_reads_prefetch for the next mm_reads_upload_packed (same two pointers, nPackedBases = P(nReads)).
 * mm_reads_packed_download returns the resident packed batch of any upload (parity tests); any pointer may be NULL.
 */
int mm_reads_upload_packed(mm_ctx* ctx, const uint32_t* bases2, const uint32_t* nmask, const uint8_t* readHasN, const int32_t* readLengths,
                           const int64_t* readStarts, size_t nReads, const int32_t* readRefGroup, const int32_t* readSelfSeqId, int32_t seqCounterBase);
int mm_reads_prefetch_packed(mm_ctx* ctx, const uint32_t* bases2, const uint32_t* nmask, size_t nPackedBases);
/*
 * One resident batch out of several packed pieces.  A reader that hands over its input in pieces of a fixed size (one page-locked buffer
 * each: skch::Map's 512 Mbp) can have several of them mapped by ONE pass -- the kernels of a pass fill the GPU only from a few Gbp
 * on -- without ever holding them in one host buffer: part p is exactly what one mm_reads_upload_packed call takes, the parts are laid
 * end to end in HBM and their reads numbered consecutively (read r of part p has readId = reads of the parts before it + r).
 * mm_reads_prefetch_packed_append is mm_reads_prefetch_packed that ADDS a piece to what has been sent ahead instead of replacing it
 * (same thread exception): the upload takes every part it finds staged from there and copies the others itself; staged pieces an upload
 * does not name stay staged for the next one.  reservePackedBases: packed bases the staging area should hold when it has to be
 * (re)allocated -- it can only grow while nothing is staged; a piece that does not fit is simply not staged: *staged (may be NULL)
 * says which happened, and a piece that was declined travels with its upload.
 * A staged piece is recognised by its two host pointers and its packed length: between the call that stages it and the upload that names
 * it (or mm_reads_prefetch_drop) the host words must neither change nor be handed to another batch -- a caller that recycles page-locked
 * buffers gives a buffer back only after the upload of its batch has returned (skch::Map does), or drops what it staged.  An ASCII upload
 * (mm_reads_upload / _device) and mm_reads_prefetch / mm_reads_prefetch_packed drop every staged piece; mm_reads_prefetch_drop does
 * nothing else.
 */
typedef struct {
  const uint32_t* bases2; const uint32_t* nmask; const uint8_t* readHasN; const int32_t* readLengths; const int64_t* readStarts;
  size_t nReads; const int32_t* readRefGroup; const int32_t* readSelfSeqId;
} mm_packed_part;
int mm_reads_upload_packed_parts(mm_ctx* ctx, const mm_packed_part* parts, size_t nParts, int32_t seqCounterBase);
int mm_reads_prefetch_packed_append(mm_ctx* ctx, const uint32_t* bases2, const uint32_t* nmask, size_t nPackedBases, size_t reservePackedBases, int* staged);
int mm_reads_prefetch_drop(mm_ctx* ctx);
/* allocates the staging area (and the copy stream) for reservePackedBases ahead of the first piece, e.g. while the index is being built:
 * the first mm_reads_prefetch_packed_append then does not stop for a multi-gigabyte hipMalloc.  Only while nothing is staged. */
int mm_reads_prefetch_reserve(mm_ctx* ctx, size_t reservePackedBases);
size_t mm_pack_read(const char* ascii, size_t len, uint32_t* bases2, uint32_t* nmask);
size_t mm_pack_read_portable(const char* ascii, size_t len, uint32_t* bases2, uint32_t* nmask);
int mm_reads_packed_download(mm_ctx* ctx, uint32_t* bases2, uint32_t* nmask, uint32_t* readHasN, size_t* nPackedBases);
/*
 * Batch slots.  A context maps its RESIDENT batch (what the last mm_reads_upload* left); beside it, it can hold MM_BATCH_SLOTS parked
 * batches.  mm_reads_exchange swaps the resident batch with the one in `slot` -- either may be empty; device pointers change hands, nothing
 * is copied -- so that several uploaded batches stay in HBM and take turns (bench.py rotates three; a caller that maps a batch twice with
 * something else in between).  Results of the previous mm_map_fragments belong to the batch that was resident then and are gone after
 * the swap (an overlapped exchange works on its own snapshot of the records and is not affected); the context's staging buffers keep
 * their sizes, so the next pass is a steady-state pass if the incoming batch fits them and is redone the sized way if not
 * (mm_pass_totals counts both).
 */
#define MM_BATCH_SLOTS 4
int mm_reads_exchange(mm_ctx* ctx, int slot);
size_t mm_num_fragments(const mm_ctx* ctx);
int mm_fragments_download(mm_ctx* ctx, mm_fragment* out);

/* a4: CommonFunc::sketchSequence for every resident fragment (commonFunc.hpp:183) */
int mm_sketch_fragments(mm_ctx* ctx);
/* raw sketches (before frequent-seed removal): out[f*sketchSize + r], counts[f]; seqId := readId+seqCounterBase */
int mm_sketch_download(mm_ctx* ctx, mm_minmer* out, uint32_t* counts);

/*
 * The whole hot path for every resident fragment: sketch (a4) -> getSeedHits (a8) ->
 * getSeedIntervalPoints (a9) -> computeL1CandidateRegions (a10) -> computeL2MappedRegions for
 * every L1 candidate (a13) -> doL2Mapping's best-first walk with its early exit and acceptance test
 * (a12, :1182-1267; k_l2_select on the integer tables of mm_set_replay_tables) -> candidate mappings (mm_mapping).
 * Equivalent of the integer part of Map::mapSingleQueryFrag (computeMap.hpp:756); only the floats of MappingResult
 * (identity, upper bound, complexity) are host work on these integers.  Results stay on the device until downloaded.
 */
int mm_map_fragments(mm_ctx* ctx);
/* How the last mm_map_fragments went: the number of times the host waited for the device inside it, and whether it was a steady-state
 * pass -- the first pass of a context sizes every staging buffer from counts it reads back stage by stage (5-7 waits); the passes behind
 * it launch against those capacities with the counts left on the device and wait once, at the end.  A pass that outgrows a buffer is
 * redone the sized way (and counted as such here).  counts (FIVE entries since MM_ABI_VERSION 2 -- the caller's array holds at least 5 --, may be NULL): L1 candidates, L2 loci, fragments whose interval
 * points went through HBM (more than the fused kernel holds; as of the last sized pass), entries reserved for the L2 streams (one per
 * index event a candidate touches: what k_l2_locate reads 16 bytes for and k_l2_sweep at most 4), fragments the fast sketch kernel handed to
 * the exact one (the hard list: fewer than sketchSize distinct survivors below the cut, or an LDS structure overflowed). */
int mm_pass_stats(const mm_ctx* ctx, uint64_t* hostSyncs, int* steady, uint64_t* counts);
/* The context's mm_map_fragments calls so far: all of them, those that went through as steady-state passes (one host wait), and the
 * steady-state attempts that outgrew a buffer and were redone the sized way.  Any pointer may be NULL. */
int mm_pass_totals(const mm_ctx* ctx, uint64_t* passes, uint64_t* steadyPasses, uint64_t* redonePasses);
int mm_result_counts(const mm_ctx* ctx, size_t* nL1, size_t* nL2);
/* any pointer may be NULL.  l1/l2 are sorted by (frag, emission order of the reference) */
int mm_results_download(mm_ctx* ctx, mm_frag_stats* stats, mm_l1_candidate* l1, mm_l2_locus* l2);
/*
 * The candidate mappings of the batch (needs the replay tables): what mapSingleQueryFrag leaves in l2Mappings for every fragment
 * before its final std::sort (computeMap.hpp:774-800), fragment-major, inside a fragment in doL2Mapping's push order.
 */
int mm_mappings_count(const mm_ctx* ctx, size_t* n);
int mm_mappings_download(mm_ctx* ctx, mm_mapping* out, size_t cap, size_t* n);
/* device address of the same records; valid until the next map call */
int mm_mappings_device(const mm_ctx* ctx, const mm_mapping** dMappings, size_t* n);
/* post-removal sketches Q.minmerTableQuery: out[f*sketchSize + r] (valid r < stats[f].sketchSize) */
int mm_query_sketch_download(mm_ctx* ctx, mm_minmer* out);
/*
 * Options.  MM_OPT_KEEP_POINTS (default 0): by default the interval points of a fragment (getSeedIntervalPoints,
 * computeMap.hpp:857) live only in LDS/registers between the seed lookup and the L1 sweep; with 1 every fragment's sorted
 * point list is also kept in HBM so that mm_points_download can return it (parity tests); with 2 the list additionally goes through the
 * interval-point pre-filter the HBM point path applies to the fragments queued for it (k_filter_points: the intervals that cannot reach
 * minimumHits are dropped before the sort -- L1 candidates unchanged), and mm_points_download returns what the filter left.
 * MM_OPT_KEEP_FULL_INDEX (default 0): keep minmerIndex as it is BEFORE dropFreqSeedSet (winSketch.hpp:497) on the host so that
 * mm_index_download_full can return it -- that is what --saveIndex writes (winSketch.hpp:127-134 run before the drop).
 * MM_OPT_RESERVE_FRAGMENTS (default 0): the number of fragments of the largest batch the caller is going to upload.  The pass that sizes the
 * context's staging buffers (the first one, or one that outgrew them) then sizes them for a batch of that many fragments -- per-fragment
 * buffers directly, count-dependent ones (candidates, L2 streams, mappings) in proportion -- so that a caller whose batches grow (skch::Map
 * ramps its device passes up from one reader batch to four) pays for sizing and allocation once.
 */
enum { MM_OPT_KEEP_POINTS = 1, MM_OPT_KEEP_FULL_INDEX = 2, MM_OPT_RESERVE_FRAGMENTS = 3 };
int mm_set_option(mm_ctx* ctx, int option, int value);
/* sorted interval points of fragment f after the seqId filters of computeMap.hpp:891-896 (needs MM_OPT_KEEP_POINTS; (seqId,pos,side) only, hash = 0) */
int mm_points_download(mm_ctx* ctx, size_t frag, mm_interval_point* out, size_t cap, size_t* n);
/* copies the L2 loci (fragment-major device order, not re-sorted) into caller-owned DEVICE memory, e.g. a torch tensor
 * that is then exchanged with RCCL; *n receives the count, cap is the capacity of dst in records */
int mm_results_copy_device(mm_ctx* ctx, mm_l2_locus* dDst, size_t cap, size_t* n);
/* device addresses of the result arrays (for RCCL all-gatherv by the caller); valid until the next map call */
int mm_results_device(const mm_ctx* ctx, const mm_l2_locus** dL2, size_t* nL2);

/*
 * a5-a7 on the device: CommonFunc::addMinmers per contig (commonFunc.hpp:302), Sketch::index
 * (winSketch.hpp:379) and the frequent-seed filter (:410-504), leaving the index resident.
 * contigOffsets has nContigs+1 entries into `bases` (ASCII).  kmerPctThreshold as Parameters::kmer_pct_threshold.
 */
int mm_index_build(mm_ctx* ctx, const char* bases, const int64_t* contigOffsets, size_t nContigs,
                   const int32_t* refGroup, float kmerPctThreshold);
/* the resident index back on the host in reference layout (sizes first, then fill) */
int mm_index_sizes(const mm_ctx* ctx, size_t* nMinmers, size_t* nKeys, size_t* nPoints, size_t* nFreq, int32_t* freqThreshold);
int mm_index_download(mm_ctx* ctx, mm_minmer* minmers, uint64_t* keys, uint64_t* offsets, mm_interval_point* points,
                      uint64_t* freqSeeds);
/* how the resident index lies in HBM: what stands in for minmerPosLookupIndex (winSketch.hpp:101) -- the
 * open-addressing seed table, whether it is the tagged layout the library picks for tables above 1 GiB (one tag byte per slot, keys
 * placed bucket-wise), the presence filter in front of it -- and for minmerIndex (:102): the merged insert / eviction event stream and
 * the per-block lists of open records.  Lets a caller (and the human-scale parity tests) see which lookup kernel variant a map call
 * will run. */
typedef struct {
  uint64_t seedTableSlots, seedTableBytes;   /* slots of 16 bytes */
  uint64_t tagBytes;                         /* 0 unless tagged */
  uint64_t filterBytes;                      /* 0: no presence filter */
  uint64_t events, openRecords;              /* entries of the event stream / of the open-record lists */
  int32_t  tagged, pad_;
} mm_index_layout;
int mm_index_layout_get(const mm_ctx* ctx, mm_index_layout* out);

/*
 * Index persistence (winSketch.hpp:284-374).  mm_index_download_full: the pre-drop minmerIndex (PREFIX.index); the lookup map of
 * mm_index_download is already the full one (PREFIX.map).  mm_index_upload_full: what --loadIndex reads -- the pre-drop
 * minmerIndex and the map, flattened as for mm_index_upload, keys in any order -- followed inside the library by
 * computeFreqHist / computeFreqSeedSet / dropFreqSeedSet (:410-504), exactly the steps the reference runs after loading.
 */
int mm_index_download_full(mm_ctx* ctx, mm_minmer* out, size_t* n);
int mm_index_upload_full(mm_ctx* ctx, const mm_minmer* minmersAll, size_t nMinmers, const uint64_t* keys, const uint64_t* offsets,
                         size_t nKeys, const mm_interval_point* points, size_t nPoints, const int32_t* contigLen,
                         const int32_t* refGroup, size_t nContigs, float kmerPctThreshold);

/*
 * Multi-GPU: reads are sharded over GPUs in contiguous blocks, the index is replicated, and the one exchange
 * step is an all-gatherv of the candidate mappings over RCCL / xGMI -- what a multi-GPU skch::Map needs before the one-to-one
 * filter (computeMap.hpp:358-405) and to write one output stream in input order.  One mm_ctx per GPU.
 *   one process per GPU:  rank 0 calls mm_comm_unique_id and ships the MM_COMM_ID_BYTES to the others (MPI, a file, torch.distributed's
 *                         store ...); every rank calls mm_comm_init_rank; after each mm_map_fragments every rank calls mm_allgatherv_mappings.
 *   one process, n GPUs:  mm_comm_init_local(ctxs, n), then mm_allgatherv_mappings_local(ctxs, n) after all n contexts have mapped their
 *                         batch (contexts that share a device exchange by device copies: RCCL wants distinct GPUs).
 * Afterwards every context holds all ranks' records, rank-major (= input order with contiguous read blocks): mm_gathered_*.
 * mm_index_replicate copies a resident index GPU to GPU (xGMI) instead of building it once per GPU; the replica serves
 * mm_map_fragments but not the mm_index_download calls (the host mirrors stay with the source context).
 */
#define MM_COMM_ID_BYTES 128
int mm_comm_unique_id(void* id);
int mm_comm_init_rank(mm_ctx* ctx, const void* id, int rank, int world);
int mm_comm_init_local(mm_ctx** ctxs, int n);
int mm_comm_world(const mm_ctx* ctx, int* rank, int* world);
/* what the communicator itself says: the number of ranks RCCL sees in it (ncclCommCount; -1 if the bound library does not export it) and
 * the file its entry points were bound from -- for a record of a multi-GPU run to show that RCCL, and which one, carried the exchange */
int mm_comm_info(const mm_ctx* ctx, int* worldSeen, char* libraryPath, size_t cap);
int mm_allgatherv_mappings(mm_ctx* ctx);
/* The same exchange overlapped with the next batch: _begin snapshots the resident candidate mappings and returns at once (the exchange
 * runs on a stream and a host thread of its own); the caller may upload and map the next batch; _end waits for the exchange, after
 * which mm_gathered_* serve the records of the batch _begin was called on.  One exchange in flight per context; every rank must call
 * _begin / _end in the same order as the other ranks' (they are collectives). */
int mm_allgatherv_mappings_begin(mm_ctx* ctx);
int mm_allgatherv_mappings_end(mm_ctx* ctx);
int mm_allgatherv_mappings_local(mm_ctx** ctxs, int n);
int mm_gathered_counts(const mm_ctx* ctx, size_t* perRank, size_t* total);
int mm_gathered_download(mm_ctx* ctx, mm_mapping* out, size_t cap);
/* device address of the gathered records.  Lifetime: valid until the next exchange of this context STARTS (mm_allgatherv_mappings,
 * mm_allgatherv_mappings_local or mm_allgatherv_mappings_begin): the exchange may reallocate the buffer and rewrites the counts, from
 * its own thread in the overlapped form -- consume (or copy) the records of batch i before calling _begin for batch i+1. */
int mm_gathered_device(const mm_ctx* ctx, const mm_mapping** dMappings, size_t* total);
int mm_index_replicate(mm_ctx* dst, mm_ctx* src);

/* per-kernel device timing, measured with hipEvents on the ctx stream (bench.py roofline leg) */
enum { MM_K_PACK = 0, MM_K_SKETCH, MM_K_SKETCH_HARD, MM_K_LOOKUP, MM_K_SORT, MM_K_L1, MM_K_L2, MM_K_REFHASH, MM_K_L2_LOCATE, MM_K_WINNOW, MM_K_SELECT, MM_K_COUNT };
int mm_profile_enable(mm_ctx* ctx, int on);
/* ms[i] = accumulated milliseconds, launches[i] = launch count since the last reset */
int mm_profile_read(mm_ctx* ctx, double* ms, uint64_t* launches, int reset);
const char* mm_kernel_name(int which);
/* integer-roofline yardstick: a kernel that does nothing but the 2 x MurmurHash3_x64_128 per position of
 * every resident fragment (both strands, same tables and staging as the sketch kernel); *msAvg = average of `reps` launches */
int mm_bench_hash_only(mm_ctx* ctx, int reps, double* msAvg);
int mm_synchronize(mm_ctx* ctx);
/* the HIP stream all work of this ctx is ordered on (a hipStream_t) */
void* mm_stream(const mm_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif

// mashmap_amd/csrc/mm_api.hip -- the extern "C" surface declared in include/mashmap_hip.h.
#include <chrono>
#include "mm_internal.h"
#include <algorithm>
#include <cstring>
#include <thread>
#include "../host/mm_stats.hpp"
#include "../host/pack2bit.hpp"

static thread_local std::string g_createErr;

static const char* kKernelNames[MM_K_COUNT] = {
  "k_pack2bit", "k_sketch_fast", "k_sketch_hard", "k_seed_lookup", "k_sort_points",
  "k_l1_sweep", "k_l2_sweep", "k_ref_hash", "k_l2_locate", "k_winnow_tiles", "k_l2_select"};

std::vector<DevBuf*> mm_ctx::allBufs() {
  DeviceIndex& I = idx;
  return {&I.evKey, &I.evAux, &I.evHash, &I.contigOff, &I.opKey, &I.opAux, &I.opHash, &I.blockOff, &I.evBlock, &I.contigBlock, &I.contigLen, &I.refGroup,
          &I.htSlots, &I.htTags, &I.filter, &I.ptKeys, &I.keys, &I.keyOff, &I.keyFreq, &dMinHits, &dCutoffs, &dAscii, &dAsciiNext, &dReadSrcOff, &dReadPackOff, &dReadLen, &dReadGroup, &dReadSelf, &dReadHasN,
          &dBases2, &dNmask, &dFrags, &dSkHash, &dSkPos, &dSkStrand, &dSkCount, &dHardList, &dCounters, &dSketchSpill, &dSketchTabs, &dQHash, &dQStrand,
          &dStats, &dPtOff, &dPts, &dPtKept, &dPtIds, &dWinFreq, &dWinExt, &dWinHeap, &dWinKeys, &dWinVals, &dWinOffH, &dWinOffT, &dWinCntH, &dWinCntT, &dL1, &dL1b, &dL1Cursors, &dL1Off, &dL1Regions, &dL2, &dL2Info, &dL2Cnt, &dL2Off, &dL2Ops, &dScanTmp, &dL2Tmp, &dL2Wide, &dL2Exact, &dL2Cells,
          &dListB, &dListC, &dBigList, &dMidList, &dL2Sort[0], &dL2Sort[1], &dL2Sort[2], &dL2Sort[3], &dL2Order, &dL2OrderPos, &dL2InitCells, &dL2InitState, &dL2First, &dL2Num, &dAccept, &dMinIsz, &dSelCnt, &dSelOff, &dSelHeap, &dFragTab, &dMappings, &dCommCounts, &dGathered};
}

extern "C" {

int mm_abi_version(void) { return MM_ABI_VERSION; }
const char* mm_kernel_name(int which) { return (which >= 0 && which < MM_K_COUNT) ? kKernelNames[which] : "?"; }

const char* mm_last_error(const mm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_createErr.c_str(); }

int mm_create(mm_ctx** out, int device, const mm_params* p) {
  if (!out || !p) { g_createErr = "mm_create: null argument"; return MM_ERR_ARG; }
  *out = nullptr;
  if (p->kmerSize < 1 || p->kmerSize > 64 || p->sketchSize < 1 || p->segLength < p->kmerSize) {
    g_createErr = "mm_create: unsupported parameters (need 1 <= kmerSize <= 64, sketchSize >= 1, segLength >= kmerSize)";
    return MM_ERR_ARG;
  }
  if (mm_check_params(p, g_createErr) != MM_OK) return MM_ERR_ARG;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    g_createErr = std::string("mm_create: no usable HIP device (") + (e == hipSuccess ? "index out of range" : hipGetErrorString(e)) + ")";
    return MM_ERR_DEVICE;
  }
  if ((e = hipSetDevice(device)) != hipSuccess) { g_createErr = hipGetErrorString(e); return MM_ERR_DEVICE; }
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { g_createErr = hipGetErrorString(e); return MM_ERR_DEVICE; }
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    g_createErr = std::string("mm_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName;
    return MM_ERR_DEVICE;
  }
  mm_ctx* c = new mm_ctx();
  c->device = device; c->P = *p;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&c->evA)) != hipSuccess || (e = hipEventCreate(&c->evB)) != hipSuccess) {
    g_createErr = hipGetErrorString(e); delete c; return MM_ERR_DEVICE;
  }
  *out = c;
  return MM_OK;
}

void mm_destroy(mm_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->gatherThread.joinable()) c->gatherThread.join();
  mm_comm_release(c);
  if (c->commStream) (void)hipStreamDestroy(c->commStream);
  if (c->copyStream) { (void)hipStreamSynchronize(c->copyStream); (void)hipStreamDestroy(c->copyStream); }
  if (c->copyDone) (void)hipEventDestroy(c->copyDone);
  for (DevBuf* b : c->allBufs()) b->release();
  c->dGatherSrc.release();
  for (auto& p : c->parked)
    for (DevBuf* b : {&p.dReadSrcOff, &p.dReadPackOff, &p.dReadLen, &p.dReadGroup, &p.dReadSelf, &p.dReadHasN, &p.dBases2, &p.dNmask, &p.dFrags}) b->release();
  if (c->hPass) (void)hipHostFree(c->hPass);
  for (auto& pr : c->evPool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (c->evA) (void)hipEventDestroy(c->evA);
  if (c->evB) (void)hipEventDestroy(c->evB);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int mm_synchronize(mm_ctx* c) { MM_HIP(c, hipStreamSynchronize(c->stream)); return MM_OK; }
void* mm_stream(const mm_ctx* c) { return (void*)c->stream; }

int mm_set_option(mm_ctx* c, int option, int value) {
  if (option == MM_OPT_KEEP_POINTS) { c->keepPoints = value != 0; c->keepFiltered = value == 2; c->ptsCap = 0; return MM_OK; }
  if (option == MM_OPT_KEEP_FULL_INDEX) { c->keepFullIndex = value != 0; return MM_OK; }
  if (option == MM_OPT_RESERVE_FRAGMENTS) { c->reserveFrags = value > 0 ? (size_t)value : 0; return MM_OK; }
  c->err = "mm_set_option: unknown option"; return MM_ERR_ARG;
}

int mm_profile_enable(mm_ctx* c, int on) { c->profile = on != 0; return MM_OK; }
int mm_profile_read(mm_ctx* c, double* ms, uint64_t* launches, int reset) {
  if (!c->evPending.empty()) { MM_HIP(c, hipSetDevice(c->device)); MM_HIP(c, hipStreamSynchronize(c->stream)); mm_profile_collect(c); }
  for (int i = 0; i < MM_K_COUNT; i++) { if (ms) ms[i] = c->kMs[i]; if (launches) launches[i] = c->kLaunches[i]; }
  if (reset) for (int i = 0; i < MM_K_COUNT; i++) { c->kMs[i] = 0; c->kLaunches[i] = 0; }
  return MM_OK;
}

// ---------------------------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------------------------
int mm_index_upload(mm_ctx* c, const mm_minmer* minmers, size_t nMinmers, const uint64_t* keys, const uint64_t* offsets,
                    size_t nKeys, const mm_interval_point* points, size_t nPoints, const uint64_t* freqSeeds, size_t nFreq,
                    const int32_t* contigLen, const int32_t* refGroup, size_t nContigs) {
  if ((nMinmers && !minmers) || (nKeys && (!keys || !offsets)) || (nPoints && !points) || (nFreq && !freqSeeds) || !contigLen || !nContigs) {
    c->err = "mm_index_upload: null argument"; return MM_ERR_ARG;
  }
  MM_HIP(c, hipSetDevice(c->device));
  c->mirrorMinmers = c->mirrorMap = false;
  c->hMinmers.assign(minmers, minmers + nMinmers);
  c->hKeys.assign(keys, keys + nKeys);
  c->hOffsets.assign(offsets, offsets + nKeys + (nKeys ? 1 : 0));
  if (!nKeys) c->hOffsets.assign(1, 0);
  c->hPoints.assign(points, points + nPoints);
  c->hFreq.assign(freqSeeds, freqSeeds + nFreq);
  std::sort(c->hFreq.begin(), c->hFreq.end());
  if (c->hOffsets.back() != nPoints) { c->err = "mm_index_upload: offsets[nKeys] != nPoints"; return MM_ERR_ARG; }
  c->mapped = false;
  return mm_build_device_index(c, contigLen, refGroup, nContigs);
}

int mm_set_tables(mm_ctx* c, const int32_t* minHits, size_t nMinHits, const int32_t* cutoffs, size_t nCutoffs) {
  if (!minHits || nMinHits < (size_t)c->P.sketchSize + 1 || !cutoffs || nCutoffs < 2) {
    c->err = "mm_set_tables: need sketchSize+1 minHits entries and the sketchCutoffs table"; return MM_ERR_ARG;
  }
  MM_HIP(c, hipSetDevice(c->device));
  MM_HIP(c, c->dMinHits.ensure(nMinHits * 4));
  MM_HIP(c, c->dCutoffs.ensure(nCutoffs * 4));
  MM_HIP(c, hipMemcpyAsync(c->dMinHits.p, minHits, nMinHits * 4, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemcpyAsync(c->dCutoffs.p, cutoffs, nCutoffs * 4, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  c->nMinHits = nMinHits; c->nCutoffs = nCutoffs;
  return MM_OK;
}

int mm_set_tables_default(mm_ctx* c, float pi) {
  const int s = c->P.sketchSize, k = c->P.kmerSize;
  std::vector<int32_t> mh = mmhost::minHitsTable(s, k, pi);
  std::vector<int> cut = mmhost::sketchCutoffs(s, k, mmhost::fixed::ANIDiff, mmhost::fixed::ANIDiffConf, (c->P.flags & MM_FLAG_HG_FILTER) != 0);
  std::vector<int32_t> cut32(cut.begin(), cut.end());
  int rc = mm_set_tables(c, mh.data(), mh.size(), cut32.data(), cut32.size());
  if (rc != MM_OK) return rc;
  std::vector<uint8_t> accept; std::vector<int16_t> minIsz;
  mmhost::replayTables(s, k, pi, mmhost::fixed::ANIDiff, true, std::max(1u, std::thread::hardware_concurrency()), accept, minIsz);
  return mm_set_replay_tables(c, accept.data(), minIsz.data(), (size_t)s + 1);
}

float mm_stat_j2md(float j, int k) { return mmhost::Stat::j2md(j, k); }
float mm_stat_md2j(float d, int k) { return mmhost::Stat::md2j(d, k); }
float mm_stat_md_lower_bound(float d, int s, int k, float ci) { return mmhost::Stat::md_lower_bound(d, s, k, ci); }
int mm_stat_min_hits_relaxed(int s, int k, float pi) { return mmhost::Stat::estimateMinimumHitsRelaxed(s, k, pi, mmhost::fixed::confidence_interval); }
int64_t mm_stat_recommended_sketch_size(int k, float pi, int64_t segLength, uint64_t referenceSize) {
  return mmhost::Stat::recommendedSketchSize(mmhost::fixed::pval_cutoff, mmhost::fixed::confidence_interval, k, 4, pi, segLength, referenceSize);
}
int mm_stat_replay_tables(int sketchSize, int k, float percentageIdentity, float ANIDiff, int keepLowPctId, uint8_t* accept, int16_t* minIsz) {
  if (sketchSize < 1 || !accept || !minIsz) return MM_ERR_ARG;
  std::vector<uint8_t> a; std::vector<int16_t> m;
  mmhost::replayTables(sketchSize, k, percentageIdentity, ANIDiff, keepLowPctId != 0, std::max(1u, std::thread::hardware_concurrency()), a, m);
  std::memcpy(accept, a.data(), a.size()); std::memcpy(minIsz, m.data(), m.size() * 2);
  return MM_OK;
}
int mm_stat_sketch_cutoffs(int sketchSize, int k, int hgFilter, int32_t* out, size_t cap) {
  std::vector<int> cut = mmhost::sketchCutoffs(sketchSize, k, mmhost::fixed::ANIDiff, mmhost::fixed::ANIDiffConf, hgFilter != 0);
  size_t n = std::min(cap, cut.size());
  for (size_t i = 0; i < n; i++) out[i] = cut[i];
  return (int)cut.size();
}

// ---------------------------------------------------------------------------------------------
// reads
// ---------------------------------------------------------------------------------------------
// fragments of one read as Map::mapModule cuts them (computeMap.hpp:587-671): pk = packed base of the read's first base
static inline void cut_read(mm_ctx* c, std::vector<DFrag>& dfr, size_t r, int32_t len, int64_t pk, int32_t& maxLen, bool& anyLong) {
  const int k = c->P.kmerSize, L = c->P.segLength;
  const bool split = !(c->P.flags & MM_FLAG_NO_SPLIT);
  if (len < k) return;                                // computeMap.hpp:325 (shorter reads are skipped)
  if (!split || len <= L) {                           // :587 -- with split off a read longer than segLength is ONE fragment (windowLen = len - segLength, :933)
    if (len > L) anyLong = true;
    c->hFrags.push_back(mm_fragment{(int32_t)r, 0, len, 0});
    dfr.push_back(DFrag{pk, len, (int32_t)r});
    maxLen = std::max(maxLen, len);
    return;
  }
  const int nfull = len / L;                          // :610
  for (int i = 0; i < nfull; i++) {
    c->hFrags.push_back(mm_fragment{(int32_t)r, i * L, L, 0});
    dfr.push_back(DFrag{pk + (int64_t)i * L, L, (int32_t)r});
  }
  if (nfull >= 1 && len % L != 0) {                   // :644
    c->hFrags.push_back(mm_fragment{(int32_t)r, len - L, L, 0});
    dfr.push_back(DFrag{pk + (int64_t)(len - L), L, (int32_t)r});
  }
  maxLen = std::max(maxLen, L);
}

// the per-read arrays, the fragment table and the run-off words behind the packed bases; ends with the stream synchronised (the host
// staging vectors go out of scope)
static int finish_upload(mm_ctx* c, size_t nReads, int64_t pk, const std::vector<int64_t>& srcOff, const std::vector<int64_t>& packOff, const std::vector<int32_t>& rlen,
                         const std::vector<int32_t>& grp, const std::vector<int32_t>& self, const std::vector<uint32_t>* hasN32, const std::vector<DFrag>& dfr, bool packOnDevice) {
  MM_HIP(c, hipMemcpyAsync(c->dReadSrcOff.p, srcOff.data(), (nReads + 1) * 8, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemcpyAsync(c->dReadPackOff.p, packOff.data(), (nReads + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (nReads) {
    MM_HIP(c, hipMemcpyAsync(c->dReadLen.p, rlen.data(), nReads * 4, hipMemcpyHostToDevice, c->stream));
    MM_HIP(c, hipMemcpyAsync(c->dReadGroup.p, grp.data(), nReads * 4, hipMemcpyHostToDevice, c->stream));
    MM_HIP(c, hipMemcpyAsync(c->dReadSelf.p, self.data(), nReads * 4, hipMemcpyHostToDevice, c->stream));
  }
  if (hasN32) MM_HIP(c, hipMemcpyAsync(c->dReadHasN.p, hasN32->data(), nReads * 4 + 4, hipMemcpyHostToDevice, c->stream));
  else MM_HIP(c, hipMemsetAsync(c->dReadHasN.p, 0, nReads * 4 + 4, c->stream));
  MM_HIP(c, hipMemsetAsync((char*)c->dBases2.p + pk / 4, 0, 64, c->stream));   // run-off words read by the last fragment
  MM_HIP(c, hipMemsetAsync((char*)c->dNmask.p + pk / 8, 0, 64, c->stream));
  if (!dfr.empty()) MM_HIP(c, hipMemcpyAsync(c->dFrags.p, dfr.data(), dfr.size() * sizeof(DFrag), hipMemcpyHostToDevice, c->stream));
  if (packOnDevice) { const int rc = mm_launch_pack(c); if (rc != MM_OK) return rc; }
  MM_HIP(c, hipStreamSynchronize(c->stream));
  return MM_OK;
}

static int ensure_read_buffers(mm_ctx* c, size_t nReads, int64_t pk, size_t nFrags) {
  // MM_OPT_RESERVE_FRAGMENTS: room for the largest announced batch right away (its fragments are full segments; a read has at least one)
  if (c->reserveFrags > nFrags) {
    const int64_t pkR = (int64_t)c->reserveFrags * (int64_t)c->P.segLength; pk = pkR > pk ? pkR + pkR / 64 : pk;
    nReads = std::max(nReads, c->reserveFrags); nFrags = c->reserveFrags;
  }
  MM_HIP(c, c->dReadSrcOff.ensure((nReads + 1) * 8)); MM_HIP(c, c->dReadPackOff.ensure((nReads + 1) * 8));
  MM_HIP(c, c->dReadLen.ensure(nReads * 4 + 4)); MM_HIP(c, c->dReadGroup.ensure(nReads * 4 + 4)); MM_HIP(c, c->dReadSelf.ensure(nReads * 4 + 4));
  MM_HIP(c, c->dReadHasN.ensure(nReads * 4 + 4));
  MM_HIP(c, c->dBases2.ensure((size_t)pk / 4 + 64)); MM_HIP(c, c->dNmask.ensure((size_t)pk / 8 + 64));
  MM_HIP(c, c->dFrags.ensure(nFrags * sizeof(DFrag) + 16));
  return MM_OK;
}

// ASCII bases (host or device memory): normalised and packed on the device (k_pack2bit)
static int upload_reads_ascii(mm_ctx* c, const void* ascii, bool onDevice, const int64_t* readOffsets, size_t nReads, const int32_t* readRefGroup,
                              const int32_t* readSelfSeqId, int32_t seqCounterBase) {
  if (!readOffsets) { c->err = "mm_reads_upload: null argument"; return MM_ERR_ARG; }
  // the prefetch state (staging buffer, its copy event, what it holds) is matched and consumed below, and the copies out of the staging
  // buffer have completed when this function returns: a mm_reads_prefetch* from another thread waits for that
  std::lock_guard<std::mutex> prefetchLock(c->prefetchMu);
  MM_HIP(c, hipSetDevice(c->device));
  std::vector<int64_t> srcOff(nReads + 1), packOff(nReads + 1);
  std::vector<int32_t> rlen(nReads);
  c->hFrags.clear();
  std::vector<DFrag> dfr;
  dfr.reserve(c->hFrags.capacity() ? c->hFrags.capacity() : nReads * 2 + 16);
  int64_t pk = 0; int32_t maxLen = 0; bool anyLong = false;
  for (size_t r = 0; r < nReads; r++) {
    const int64_t len64 = readOffsets[r + 1] - readOffsets[r];
    if (len64 < 0 || len64 > 0x7fffffff) { c->err = "mm_reads_upload: read length out of range (offset_t is int32, base_types.hpp:21)"; return MM_ERR_ARG; }
    srcOff[r] = readOffsets[r]; packOff[r] = pk; rlen[r] = (int32_t)len64;
    cut_read(c, dfr, r, (int32_t)len64, pk, maxLen, anyLong);
    pk += (len64 + 31) / 32 * 32;
  }
  srcOff[nReads] = readOffsets[nReads]; packOff[nReads] = pk;
  c->nReads = nReads; c->nFrags = dfr.size(); c->nPackedBases = (size_t)pk; c->seqCounterBase = seqCounterBase; c->maxFragLen = maxLen;
  c->sketched = false; c->mapped = false; c->fragTabStale = true; c->gathered = false;
  c->windowed = anyLong;
  const size_t srcBase = (size_t)readOffsets[0];
  const size_t nSrc = (size_t)(readOffsets[nReads] - readOffsets[0]);
  if (nSrc && !ascii) { c->err = "mm_reads_upload: null argument"; return MM_ERR_ARG; }
  for (size_t r = 0; r <= nReads; r++) srcOff[r] -= (int64_t)srcBase;
  // bytes mm_reads_prefetch has already sent (same host range): take that buffer and wait for its copy on the device
  const bool prefetched = c->prefetchValid && !onDevice && nSrc && c->prefetchPtr == (const void*)((const char*)ascii + srcBase) && c->prefetchBytes == nSrc;
  c->prefetchValid = false;
  c->staged.clear(); c->stagedBytes = 0;             // an ASCII upload drops whatever packed pieces were sent ahead (header contract): nothing stale can match later
  if (prefetched) { std::swap(c->dAscii, c->dAsciiNext); MM_HIP(c, hipStreamWaitEvent(c->stream, c->copyDone, 0)); }
  else MM_HIP(c, c->dAscii.ensure(nSrc + 64));
  { const int rc = ensure_read_buffers(c, nReads, pk, dfr.size()); if (rc != MM_OK) return rc; }
  if (nSrc && !prefetched) MM_HIP(c, hipMemcpyAsync(c->dAscii.p, (const char*)ascii + srcBase, nSrc, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  std::vector<int32_t> grp(nReads, -1), self(nReads, -1);
  if (readRefGroup) grp.assign(readRefGroup, readRefGroup + nReads);
  if (readSelfSeqId) self.assign(readSelfSeqId, readSelfSeqId + nReads);
  return finish_upload(c, nReads, pk, srcOff, packOff, rlen, grp, self, nullptr, dfr, true);
}

// Packed pieces laid end to end: part p owns the packed bases [base_p, base_p + P_p) of the resident batch
static int upload_packed_parts(mm_ctx* c, const mm_packed_part* parts, size_t nParts, int32_t seqCounterBase) {
  if (nParts && !parts) { c->err = "mm_reads_upload_packed: null argument"; return MM_ERR_ARG; }
  std::lock_guard<std::mutex> prefetchLock(c->prefetchMu);
  MM_HIP(c, hipSetDevice(c->device));
  size_t nReads = 0;
  for (size_t p = 0; p < nParts; p++) {
    if (parts[p].nReads && !parts[p].readLengths) { c->err = "mm_reads_upload_packed: null argument"; return MM_ERR_ARG; }
    nReads += parts[p].nReads;
  }
  if (nReads > 0x7fffffff) { c->err = "mm_reads_upload_packed: more than 2^31 reads in one batch"; return MM_ERR_ARG; }
  std::vector<int64_t> srcOff(nReads + 1, 0), packOff(nReads + 1);
  std::vector<int32_t> rlen(nReads), grp(nReads, -1), self(nReads, -1);
  std::vector<uint32_t> hasN32(nReads + 1, 0u);
  std::vector<int64_t> partBase(nParts + 1, 0);
  c->hFrags.clear();
  std::vector<DFrag> dfr;
  dfr.reserve(c->hFrags.capacity() ? c->hFrags.capacity() : nReads * 2 + 16);
  int64_t pk = 0; int32_t maxLen = 0; bool anyLong = false;
  size_t r = 0;
  for (size_t p = 0; p < nParts; p++) {
    const mm_packed_part& P = parts[p];
    partBase[p] = pk;
    int64_t at = 0;                                   // packed base inside the part
    for (size_t i = 0; i < P.nReads; i++, r++) {
      const int64_t len64 = (int64_t)P.readLengths[i];
      if (len64 < 0) { c->err = "mm_reads_upload_packed: negative read length"; return MM_ERR_ARG; }
      if (P.readStarts) {                             // reads placed by the caller: gaps between them are allowed (and never read as bases)
        const int64_t want = P.readStarts[i] - P.readStarts[0];
        if (want < at || (want & 31)) { c->err = "mm_reads_upload_packed: readStarts must be ascending multiples of 32 that leave room for every read"; return MM_ERR_ARG; }
        at = want;
      }
      packOff[r] = pk + at; rlen[r] = (int32_t)len64;
      if (P.readRefGroup) grp[r] = P.readRefGroup[i];
      if (P.readSelfSeqId) self[r] = P.readSelfSeqId[i];
      if (P.readHasN) hasN32[r] = P.readHasN[i] ? 1u : 0u;
      else if (len64) {                               // the read's own mask words only: the words of a gap behind it are never read (header contract)
        if (!P.nmask) { c->err = "mm_reads_upload_packed: null argument"; return MM_ERR_ARG; }
        const uint32_t* w = P.nmask + at / 32; const size_t nw = ((size_t)len64 + 31) / 32;
        uint32_t any = 0; for (size_t j = 0; j < nw; j++) any |= w[j];
        hasN32[r] = any ? 1u : 0u;
      }
      cut_read(c, dfr, r, (int32_t)len64, pk + at, maxLen, anyLong);
      at += (len64 + 31) / 32 * 32;
    }
    if (at && (!P.bases2 || !P.nmask)) { c->err = "mm_reads_upload_packed: null argument"; return MM_ERR_ARG; }
    pk += at;
  }
  partBase[nParts] = pk; packOff[nReads] = pk;
  c->nReads = nReads; c->nFrags = dfr.size(); c->nPackedBases = (size_t)pk; c->seqCounterBase = seqCounterBase; c->maxFragLen = maxLen;
  c->sketched = false; c->mapped = false; c->fragTabStale = true; c->gathered = false;
  c->windowed = anyLong;
  c->prefetchValid = false;                           // (an ASCII prefetch, if any, is not for this upload)
  { const int rc = ensure_read_buffers(c, nReads, pk, dfr.size()); if (rc != MM_OK) return rc; }
  // the words are the device layout already: straight into dBases2 / dNmask at the part's base -- from the staging area where the piece
  // travelled ahead (mm_reads_prefetch_packed[_append]: same two pointers, same packed length), from the host otherwise
  bool waited = false;
  std::vector<char> used(c->staged.size(), 0);
  for (size_t p = 0; p < nParts; p++) {
    const int64_t n = partBase[p + 1] - partBase[p];
    if (!n) continue;
    char* dB = (char*)c->dBases2.p + partBase[p] / 4; char* dM = (char*)c->dNmask.p + partBase[p] / 8;
    size_t hit = c->staged.size();
    for (size_t q = 0; q < c->staged.size(); q++)
      if (!used[q] && c->staged[q].b2 == (const void*)parts[p].bases2 && c->staged[q].nm == (const void*)parts[p].nmask && c->staged[q].nPacked == (size_t)n) { hit = q; break; }
    if (hit < c->staged.size()) {
      if (!waited) { MM_HIP(c, hipStreamWaitEvent(c->stream, c->copyDone, 0)); waited = true; }
      used[hit] = 1;
      const char* src = (const char*)c->dAsciiNext.p + c->staged[hit].off;
      MM_HIP(c, hipMemcpyAsync(dB, src, (size_t)n / 4, hipMemcpyDeviceToDevice, c->stream));
      MM_HIP(c, hipMemcpyAsync(dM, src + (size_t)n / 4, (size_t)n / 8, hipMemcpyDeviceToDevice, c->stream));
    } else {
      MM_HIP(c, hipMemcpyAsync(dB, parts[p].bases2, (size_t)n / 4, hipMemcpyHostToDevice, c->stream));
      MM_HIP(c, hipMemcpyAsync(dM, parts[p].nmask, (size_t)n / 8, hipMemcpyHostToDevice, c->stream));
    }
  }
  const int rc = finish_upload(c, nReads, pk, srcOff, packOff, rlen, grp, self, &hasN32, dfr, false);   // synchronises: the copies out of the staging area are done
  // pieces this upload did not name stay staged; the area starts over once it is empty
  std::vector<mm_ctx::StagedPart> left;
  for (size_t q = 0; q < c->staged.size(); q++) if (!used[q]) left.push_back(c->staged[q]);
  c->staged.swap(left);
  return rc;
}

int mm_reads_upload(mm_ctx* c, const char* bases, const int64_t* readOffsets, size_t nReads, const int32_t* g, const int32_t* s, int32_t base) {
  return upload_reads_ascii(c, bases, false, readOffsets, nReads, g, s, base);
}
int mm_reads_upload_device(mm_ctx* c, const void* dBases, size_t nBases, const int64_t* readOffsets, size_t nReads, const int32_t* g,
                           const int32_t* s, int32_t base) {
  (void)nBases;
  return upload_reads_ascii(c, dBases, true, readOffsets, nReads, g, s, base);
}
int mm_reads_upload_packed(mm_ctx* c, const uint32_t* bases2, const uint32_t* nmask, const uint8_t* readHasN, const int32_t* readLengths, const int64_t* readStarts,
                           size_t nReads, const int32_t* g, const int32_t* s, int32_t base) {
  static const int32_t none = 0;
  mm_packed_part P{bases2, nmask, readHasN, readLengths ? readLengths : &none, nReads ? readStarts : nullptr, nReads, g, s};
  return upload_packed_parts(c, &P, 1, base);
}
int mm_reads_upload_packed_parts(mm_ctx* c, const mm_packed_part* parts, size_t nParts, int32_t base) { return upload_packed_parts(c, parts, nParts, base); }
size_t mm_pack_read(const char* ascii, size_t len, uint32_t* bases2, uint32_t* nmask) { return mmhost::pack2bit(ascii, len, bases2, nmask); }
size_t mm_pack_read_portable(const char* ascii, size_t len, uint32_t* bases2, uint32_t* nmask) { return mmhost::pack2bit_scalar(ascii, len, bases2, nmask); }

int mm_reads_packed_download(mm_ctx* c, uint32_t* bases2, uint32_t* nmask, uint32_t* readHasN, size_t* nPackedBases) {
  if (nPackedBases) *nPackedBases = c->nPackedBases;
  MM_HIP(c, hipSetDevice(c->device));
  if (bases2 && c->nPackedBases) MM_HIP(c, hipMemcpyAsync(bases2, c->dBases2.p, c->nPackedBases / 4, hipMemcpyDeviceToHost, c->stream));
  if (nmask && c->nPackedBases) MM_HIP(c, hipMemcpyAsync(nmask, c->dNmask.p, c->nPackedBases / 8, hipMemcpyDeviceToHost, c->stream));
  if (readHasN && c->nReads) MM_HIP(c, hipMemcpyAsync(readHasN, c->dReadHasN.p, c->nReads * 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  return MM_OK;
}

static int prefetch_streams(mm_ctx* c) {
  MM_HIP(c, hipSetDevice(c->device));
  if (!c->copyStream) MM_HIP(c, hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
  if (!c->copyDone) MM_HIP(c, hipEventCreateWithFlags(&c->copyDone, hipEventDisableTiming));
  return MM_OK;
}

int mm_reads_prefetch(mm_ctx* c, const char* bases, size_t nBytes) {
  std::lock_guard<std::mutex> prefetchLock(c->prefetchMu);
  c->prefetchValid = false;
  if (!bases || !nBytes) return MM_OK;
  { const int rc = prefetch_streams(c); if (rc != MM_OK) return rc; }
  c->staged.clear(); c->stagedBytes = 0;              // the staging area is one buffer: ASCII ahead replaces packed pieces ahead
  MM_HIP(c, c->dAsciiNext.ensure(nBytes + 64));
  MM_HIP(c, hipMemcpyAsync(c->dAsciiNext.p, bases, nBytes, hipMemcpyHostToDevice, c->copyStream));
  MM_HIP(c, hipEventRecord(c->copyDone, c->copyStream));
  c->prefetchPtr = bases; c->prefetchPtr2 = nullptr; c->prefetchBytes = nBytes; c->prefetchPacked = false; c->prefetchValid = true;
  return MM_OK;
}

// adds one packed piece to the staging area (prefetchMu held)
static int stage_packed(mm_ctx* c, const uint32_t* bases2, const uint32_t* nmask, size_t nPackedBases, size_t reservePackedBases, int* stagedOut = nullptr) {
  if (stagedOut) *stagedOut = 0;
  if (nPackedBases % 32) { c->err = "mm_reads_prefetch_packed: the packed length of a batch is a multiple of 32 bases"; return MM_ERR_ARG; }
  { const int rc = prefetch_streams(c); if (rc != MM_OK) return rc; }
  c->prefetchValid = false;                           // (packed pieces replace an ASCII batch sent ahead)
  const size_t bytes = nPackedBases / 4 + nPackedBases / 8;
  if (c->staged.empty()) {                            // nothing on its way: the area may grow, and starts over
    const size_t want = std::max(bytes, reservePackedBases / 4 + reservePackedBases / 8) + 64;
    MM_HIP(c, c->dAsciiNext.ensure(want));
    c->stagedBytes = 0;
  }
  // the area is used as a ring: pieces leave in the order they came (an upload takes the pieces of the oldest batches), so the next
  // piece goes behind the newest one, or to the front again once the oldest ones have left
  auto fits = [&](size_t off) {
    if (off + bytes + 64 > c->dAsciiNext.bytes) return false;
    for (const auto& q : c->staged) { const size_t qb = q.nPacked / 4 + q.nPacked / 8; if (off < q.off + qb && q.off < off + bytes) return false; }
    return true;
  };
  if (!fits(c->stagedBytes)) {
    if (!fits(0)) return MM_OK;                       // no room under the pieces that are on their way: this one travels with its upload
    c->stagedBytes = 0;
  }
  char* dst = (char*)c->dAsciiNext.p + c->stagedBytes;
  MM_HIP(c, hipMemcpyAsync(dst, bases2, nPackedBases / 4, hipMemcpyHostToDevice, c->copyStream));
  MM_HIP(c, hipMemcpyAsync(dst + nPackedBases / 4, nmask, nPackedBases / 8, hipMemcpyHostToDevice, c->copyStream));
  MM_HIP(c, hipEventRecord(c->copyDone, c->copyStream));   // the event always marks the last piece: an upload that waits for it has them all
  c->staged.push_back(mm_ctx::StagedPart{bases2, nmask, nPackedBases, c->stagedBytes});
  c->stagedBytes += bytes;                            // where the next piece goes
  if (stagedOut) *stagedOut = 1;
  return MM_OK;
}

int mm_reads_prefetch_packed(mm_ctx* c, const uint32_t* bases2, const uint32_t* nmask, size_t nPackedBases) {
  std::lock_guard<std::mutex> prefetchLock(c->prefetchMu);
  c->prefetchValid = false;
  c->staged.clear(); c->stagedBytes = 0;
  if (!bases2 || !nmask || !nPackedBases) return MM_OK;
  return stage_packed(c, bases2, nmask, nPackedBases, nPackedBases);
}

int mm_reads_prefetch_packed_append(mm_ctx* c, const uint32_t* bases2, const uint32_t* nmask, size_t nPackedBases, size_t reservePackedBases, int* staged) {
  std::lock_guard<std::mutex> prefetchLock(c->prefetchMu);
  if (staged) *staged = 0;
  if (!bases2 || !nmask || !nPackedBases) return MM_OK;
  return stage_packed(c, bases2, nmask, nPackedBases, reservePackedBases, staged);
}

int mm_reads_prefetch_reserve(mm_ctx* c, size_t reservePackedBases) {
  std::lock_guard<std::mutex> prefetchLock(c->prefetchMu);
  if (!c->staged.empty() || c->prefetchValid) return MM_OK;
  { const int rc = prefetch_streams(c); if (rc != MM_OK) return rc; }
  MM_HIP(c, c->dAsciiNext.ensure(reservePackedBases / 4 + reservePackedBases / 8 + 64));
  return MM_OK;
}

int mm_reads_prefetch_drop(mm_ctx* c) {
  std::lock_guard<std::mutex> prefetchLock(c->prefetchMu);
  c->prefetchValid = false;
  c->staged.clear(); c->stagedBytes = 0;
  return MM_OK;
}

void* mm_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void mm_host_free(void* p) { if (p) (void)hipHostFree(p); }

int mm_reads_exchange(mm_ctx* c, int slot) {
  if (slot < 0 || slot >= MM_BATCH_SLOTS) { c->err = "mm_reads_exchange: slot out of range"; return MM_ERR_ARG; }
  std::lock_guard<std::mutex> prefetchLock(c->prefetchMu);   // an upload of the resident batch is not under way
  MM_HIP(c, hipSetDevice(c->device));
  MM_HIP(c, hipStreamSynchronize(c->stream));                // nothing queued still reads the outgoing batch
  mm_ctx::ParkedReads& p = c->parked[slot];
  std::swap(c->nReads, p.nReads); std::swap(c->nFrags, p.nFrags); std::swap(c->nPackedBases, p.nPackedBases);
  std::swap(c->seqCounterBase, p.seqCounterBase); std::swap(c->maxFragLen, p.maxFragLen); std::swap(c->windowed, p.windowed);
  std::swap(c->dReadSrcOff, p.dReadSrcOff); std::swap(c->dReadPackOff, p.dReadPackOff); std::swap(c->dReadLen, p.dReadLen);
  std::swap(c->dReadGroup, p.dReadGroup); std::swap(c->dReadSelf, p.dReadSelf); std::swap(c->dReadHasN, p.dReadHasN);
  std::swap(c->dBases2, p.dBases2); std::swap(c->dNmask, p.dNmask); std::swap(c->dFrags, p.dFrags);
  c->hFrags.swap(p.hFrags);
  c->sketched = false; c->mapped = false; c->fragTabStale = true; c->gathered = false;
  return MM_OK;
}

size_t mm_num_fragments(const mm_ctx* c) { return c->nFrags; }
int mm_fragments_download(mm_ctx* c, mm_fragment* out) {
  if (c->nFrags) std::memcpy(out, c->hFrags.data(), c->nFrags * sizeof(mm_fragment));
  return MM_OK;
}

// ---------------------------------------------------------------------------------------------
// sketch
// ---------------------------------------------------------------------------------------------
int mm_sketch_fragments(mm_ctx* c) {
  MM_HIP(c, hipSetDevice(c->device));
  int rc = mm_launch_sketch(c);
  if (rc != MM_OK) return rc;
  MM_HIP(c, hipStreamSynchronize(c->stream));
  if (c->profile) mm_profile_collect(c);
  c->sketched = true;
  return MM_OK;
}

int mm_sketch_download(mm_ctx* c, mm_minmer* out, uint32_t* counts) {
  if (!c->sketched) { c->err = "mm_sketch_download: no sketches resident"; return MM_ERR_STATE; }
  MM_HIP(c, hipSetDevice(c->device));
  const size_t nF = c->nFrags, s = (size_t)c->P.sketchSize;
  std::vector<uint64_t> h(nF * s); std::vector<int2> p(nF * s); std::vector<int8_t> st(nF * s); std::vector<uint32_t> cnt(nF);
  if (nF) {
    MM_HIP(c, hipMemcpyAsync(h.data(), c->dSkHash.p, nF * s * 8, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(p.data(), c->dSkPos.p, nF * s * 8, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(st.data(), c->dSkStrand.p, nF * s, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(cnt.data(), c->dSkCount.p, nF * 4, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
  }
  for (size_t f = 0; f < nF; f++) {
    if (counts) counts[f] = cnt[f];
    if (!out) continue;
    for (uint32_t r = 0; r < cnt[f]; r++) {
      const size_t o = f * s + r;
      out[o] = mm_minmer{h[o], p[o].x, p[o].y, c->hFrags[f].readId + c->seqCounterBase, (int16_t)st[o], 0};
    }
  }
  return MM_OK;
}

// ---------------------------------------------------------------------------------------------
// map
// ---------------------------------------------------------------------------------------------
int mm_map_fragments(mm_ctx* c) {
  if (!c->idx.ready) { c->err = "mm_map_fragments: no index resident (mm_index_upload / mm_index_build first)"; return MM_ERR_STATE; }
  if (!c->nMinHits) { c->err = "mm_map_fragments: mm_set_tables first"; return MM_ERR_STATE; }
  MM_HIP(c, hipSetDevice(c->device));
  static const bool timing = getenv("MASHMAP_HIP_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  const double a0 = g_mmAllocSeconds;
  int rc = mm_launch_sketch(c);
  if (rc != MM_OK) return rc;
  c->sketched = true;
  const auto t1 = std::chrono::steady_clock::now();
  rc = mm_launch_map(c);
  if (rc != MM_OK) return rc;
  if (!c->lastSteady) { MM_HIP(c, hipStreamSynchronize(c->stream)); c->nSyncs++; }   // a steady-state pass ends with its one synchronisation
  if (timing && !c->lastSteady) {
    auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    fprintf(stderr, "[mashmap_hip::timing] sized pass (%zu fragments, %llu host waits): sketch launch %.4f s, lookup .. selection %.4f s, of both in hipMalloc / hipFree %.4f s\n",
            c->nFrags, (unsigned long long)c->nSyncs, sec(t0, t1), sec(t1, std::chrono::steady_clock::now()), g_mmAllocSeconds - a0);
  }
  if (c->profile) mm_profile_collect(c);
  c->mapped = true;
  c->nPasses++; if (c->lastSteady) c->nSteadyPasses++;
  return MM_OK;
}

int mm_pass_totals(const mm_ctx* c, uint64_t* passes, uint64_t* steadyPasses, uint64_t* redonePasses) {
  if (passes) *passes = c->nPasses;
  if (steadyPasses) *steadyPasses = c->nSteadyPasses;
  if (redonePasses) *redonePasses = c->nRedone;
  return MM_OK;
}

int mm_pass_stats(const mm_ctx* c, uint64_t* hostSyncs, int* steady, uint64_t* counts) {
  if (!c->mapped) return MM_ERR_STATE;
  if (hostSyncs) *hostSyncs = c->nSyncs;
  if (steady) *steady = c->lastSteady ? 1 : 0;
  if (counts) { counts[0] = c->nL1; counts[1] = c->nL2; counts[2] = c->lastBig; counts[3] = c->lastOps; counts[4] = c->lastHard; }
  return MM_OK;
}

int mm_result_counts(const mm_ctx* c, size_t* nL1, size_t* nL2) {
  if (!c->mapped) return MM_ERR_STATE;
  if (nL1) *nL1 = c->nL1;
  if (nL2) *nL2 = c->nL2;
  return MM_OK;
}

int mm_results_device(const mm_ctx* c, const mm_l2_locus** dL2, size_t* nL2) {
  if (!c->mapped) return MM_ERR_STATE;
  if (dL2) *dL2 = c->dL2.as<mm_l2_locus>();
  if (nL2) *nL2 = c->nL2;
  return MM_OK;
}

int mm_results_copy_device(mm_ctx* c, mm_l2_locus* dDst, size_t cap, size_t* n) {
  if (!c->mapped) { c->err = "mm_results_copy_device: nothing mapped"; return MM_ERR_STATE; }
  if (n) *n = c->nL2;
  if (c->nL2 > cap) { c->err = "mm_results_copy_device: destination too small"; return MM_ERR_ARG; }
  if (c->nL2) MM_HIP(c, hipMemcpyAsync(dDst, c->dL2.p, c->nL2 * sizeof(mm_l2_locus), hipMemcpyDeviceToDevice, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  return MM_OK;
}

int mm_index_sizes(const mm_ctx* c, size_t* nMinmers, size_t* nKeys, size_t* nPoints, size_t* nFreq, int32_t* freqThreshold) {
  if (!c->idx.ready) return MM_ERR_STATE;
  if (nMinmers) *nMinmers = c->idx.nRec;
  if (nKeys) *nKeys = c->idx.nKeys;
  if (nPoints) *nPoints = c->idx.nPoints;
  if (nFreq) *nFreq = c->hFreq.size();
  if (freqThreshold) *freqThreshold = c->freqThreshold;
  return MM_OK;
}

int mm_index_layout_get(const mm_ctx* c, mm_index_layout* o) {
  if (!c->idx.ready || !o) return MM_ERR_STATE;
  const DeviceIndex& I = c->idx;
  o->seedTableSlots = I.htCap; o->seedTableBytes = (uint64_t)I.htCap * 16;
  o->tagged = I.tagged ? 1 : 0; o->pad_ = 0;
  o->tagBytes = I.tagged ? (uint64_t)I.htCap : 0;
  o->filterBytes = I.filterMask ? (I.filterMask + 1) * 8 : 0;
  o->events = 2 * (uint64_t)I.nRec; o->openRecords = I.nOpen;
  return MM_OK;
}

int mm_index_download(mm_ctx* c, mm_minmer* minmers, uint64_t* keys, uint64_t* offsets, mm_interval_point* points, uint64_t* freqSeeds) {
  if (!c->idx.ready) { c->err = "mm_index_download: no index resident"; return MM_ERR_STATE; }
  if (c->idx.keys.bytes == 0 && c->idx.nKeys) { c->err = "mm_index_download: this context holds a replica (mm_index_replicate); ask the context the index was built on"; return MM_ERR_STATE; }
  MM_HIP(c, hipSetDevice(c->device));
  if (minmers) { const int rc = mm_mirror_minmers(c); if (rc != MM_OK) return rc; }
  if (keys || offsets || points) { const int rc = mm_mirror_map(c); if (rc != MM_OK) return rc; }
  if (minmers && !c->hMinmers.empty()) std::memcpy(minmers, c->hMinmers.data(), c->hMinmers.size() * sizeof(mm_minmer));
  if (keys && !c->hKeys.empty()) std::memcpy(keys, c->hKeys.data(), c->hKeys.size() * 8);
  if (offsets) std::memcpy(offsets, c->hOffsets.data(), c->hOffsets.size() * 8);
  if (points && !c->hPoints.empty()) std::memcpy(points, c->hPoints.data(), c->hPoints.size() * sizeof(mm_interval_point));
  if (freqSeeds && !c->hFreq.empty()) std::memcpy(freqSeeds, c->hFreq.data(), c->hFreq.size() * 8);
  return MM_OK;
}

}  // extern "C"

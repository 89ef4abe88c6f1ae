// mashmap_amd/csrc/mm_select.hip -- doL2Mapping's best-first walk over a fragment's L1 candidates on the device (gfx950).
//
//   Map::mapSingleQueryFrag   group loop + std::make_heap            src/map/include/computeMap.hpp:774-796
//   Map::doL2Mapping          best-first, ANI cut-off, acceptance    src/map/include/computeMap.hpp:1182-1267
//
// The reference computes the L2 loci of a candidate only when the walk reaches it; here k_l2_sweep has already produced the loci of
// every candidate, and this kernel decides -- in integers -- which of them doL2Mapping would have reported, in the order it would
// have pushed them: the candidate mappings that chaining and the plane-sweep filters then work on (and that multi-GPU runs
// exchange, mm_comm.hip).  The two float decisions of the walk are functions of small integers and arrive as host tables
// (mmhost::replayTables): accept[Qs][shared] (:1221) and minIsz[Qs][best] (:1192-1202).  The heap is libstdc++'s (mm_heap.h).
// One thread per fragment (a fragment has ~1 candidate; the heap lives in a global scratch slice the size of its candidate list).
// Two passes: count, exclusive scan, write -- so the records come out fragment-major without a sort.
#include "mm_internal.h"
#include "mm_select_core.h"
#include <algorithm>

struct SelectArgs {
  int nFrags, stride, hg, skipPrefix, seqCounterBase;
  const mm_fragment* fragTab;
  const mm_frag_stats* stats; const int64_t* l1Off; const mm_l1_candidate* l1;
  const int64_t* l2First; const int32_t* l2Num; const mm_l2_locus* l2;
  const int32_t* refGroup; const uint8_t* accept; const int16_t* minIsz;
  int32_t* heap;
};

template <bool WRITE>
__global__ void __launch_bounds__(256)
k_l2_select(SelectArgs A, int32_t* __restrict__ counts, const int64_t* __restrict__ outOff, mm_mapping* __restrict__ out,
            const int64_t* __restrict__ totalDev, long long outCap, unsigned long long* __restrict__ result /* steady-state passes: [0] = total, [1] = 1 if it exceeds outCap */,
            const unsigned long long* __restrict__ passCnt /* steady-state passes: the pass's counters; any overflow flag = the stages before this one are incomplete */) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (passCnt && (passCnt[1] | passCnt[3] | passCnt[5] | passCnt[6])) {     // the pass will be redone with the host's sizing: nothing here may be trusted (or dereferenced)
    if (!WRITE && f < A.nFrags) counts[f] = 0;
    return;
  }
  if (WRITE && totalDev) {
    const long long total = (long long)*totalDev;
    if (f == 0) { result[0] = (unsigned long long)total; if (total > outCap) result[1] = 1ull; }
    if (total > outCap) return;                              // the records do not fit the buffer as it is: the pass is redone with the host's sizing
  }
  if (f >= A.nFrags) return;
  const mm_frag_stats st = A.stats[f];
  const int Qs = st.sketchSize, nC = st.nL1;
  int n = 0;
  if (Qs > 0 && nC > 0) {
    const int64_t b = A.l1Off[f];
    const mm_l1_candidate* cl = A.l1 + b;
    int32_t* heap = A.heap + b;
    const uint8_t* acc = A.accept + (size_t)Qs * A.stride;
    const int16_t* cut = A.minIsz + (size_t)Qs * A.stride;
    mm_mapping rec;
    if (WRITE) {
      const mm_fragment fr = A.fragTab[f];
      rec.querySeqId = A.seqCounterBase + fr.readId; rec.fragStart = fr.fragStart; rec.fragLen = fr.len;
      rec.sketchSize = Qs; rec.rawSketchSize = st.rawSketchSize; rec.pad_ = 0; rec.maxHash = st.maxHash;
    }
    mm_mapping* dst = WRITE ? out + outOff[f] : nullptr;
    int w = 0;
    n = mm_select_fragment(nC, cl, heap, A.l2First + b, A.l2Num + b, A.l2, A.refGroup, A.skipPrefix, A.hg, Qs, acc, cut, [&](const mm_l2_locus& L) {
      if (WRITE) { rec.refSeqId = L.seqId; rec.refStartPos = L.meanOptimalPos; rec.conservedSketches = L.sharedSketchSize; rec.strand = L.strand; dst[w++] = rec; }
    });
  }
  if (!WRITE) counts[f] = n;
}

int mm_scan_i32_to_i64(mm_ctx* c, int64_t n, const int32_t* dIn, int64_t* dOut, int64_t* total);   // mm_l2.hip

int mm_launch_select(mm_ctx* c, bool steady) {
  c->nMappings = 0;
  const int nF = (int)c->nFrags;
  if (!c->haveReplayTables || nF == 0 || (!steady && c->nL1 == 0)) return MM_OK;
  const size_t cF = mm_frag_cap(c, (size_t)nF);
  MM_HIP(c, c->dSelCnt.ensure(cF * 4 + 64)); MM_HIP(c, c->dSelOff.ensure(cF * 8 + 64));
  MM_HIP(c, c->dSelHeap.ensure((steady ? c->candCap : std::max(c->nL1, c->candCap)) * 4 + 64));
  MM_HIP(c, c->dFragTab.ensure(cF * sizeof(mm_fragment) + 64));
  if (c->fragTabStale) {
    MM_HIP(c, hipMemcpyAsync(c->dFragTab.p, c->hFrags.data(), (size_t)nF * sizeof(mm_fragment), hipMemcpyHostToDevice, c->stream));
    c->fragTabStale = false;
  }
  SelectArgs A;
  A.nFrags = nF; A.stride = c->P.sketchSize + 1; A.hg = (c->P.flags & MM_FLAG_HG_FILTER) ? 1 : 0; A.skipPrefix = (c->P.flags & MM_FLAG_SKIP_PREFIX) ? 1 : 0;
  A.seqCounterBase = c->seqCounterBase;
  A.fragTab = c->dFragTab.as<mm_fragment>();
  A.stats = c->dStats.as<mm_frag_stats>(); A.l1Off = c->dL1Off.as<int64_t>(); A.l1 = c->dL1.as<mm_l1_candidate>();
  A.l2First = c->dL2First.as<int64_t>(); A.l2Num = c->dL2Num.as<int32_t>(); A.l2 = c->dL2.as<mm_l2_locus>();
  A.refGroup = c->idx.refGroup.as<int32_t>(); A.accept = c->dAccept.as<uint8_t>(); A.minIsz = c->dMinIsz.as<int16_t>();
  A.heap = c->dSelHeap.as<int32_t>();
  KernelTimer t(c, MM_K_SELECT);
  hipLaunchKernelGGL((k_l2_select<false>), dim3((nF + 255) / 256), dim3(256), 0, c->stream, A, c->dSelCnt.as<int32_t>(), (const int64_t*)nullptr, (mm_mapping*)nullptr,
                     (const int64_t*)nullptr, 0ll, (unsigned long long*)nullptr, steady ? (const unsigned long long*)(c->dCounters.as<unsigned long long>() + 8) : (const unsigned long long*)nullptr);
  MM_HIP(c, hipGetLastError());
  if (steady) {
    // the records' number stays on the device: the writing pass checks it against the buffer as the previous pass left it and
    // reports both in the counters the launcher reads when the pass is over (dCounters[32], [33])
    const int64_t* dTotal = nullptr;
    const int rc = mm_scan_i32_to_i64_dev(c, nF, c->dSelCnt.as<int32_t>(), c->dSelOff.as<int64_t>(), &dTotal);
    if (rc != MM_OK) return rc;
    const long long cap = (long long)(c->dMappings.bytes / sizeof(mm_mapping)) - 2;
    hipLaunchKernelGGL((k_l2_select<true>), dim3((nF + 255) / 256), dim3(256), 0, c->stream, A, (int32_t*)nullptr, c->dSelOff.as<int64_t>(), c->dMappings.as<mm_mapping>(),
                       dTotal, cap, c->dCounters.as<unsigned long long>() + 32, (const unsigned long long*)(c->dCounters.as<unsigned long long>() + 8));
    MM_HIP(c, hipGetLastError());
    return MM_OK;
  }
  int64_t total = 0;
  const int rc = mm_scan_i32_to_i64(c, nF, c->dSelCnt.as<int32_t>(), c->dSelOff.as<int64_t>(), &total);
  c->nSyncs++;
  if (rc != MM_OK) return rc;
  { const size_t tot = mm_scaled(c, (size_t)total, sizeof(mm_mapping)); MM_HIP(c, c->dMappings.ensure((tot + tot / 16) * sizeof(mm_mapping) + 4096)); }   // head room for the steady-state passes behind this one
  if (total) {
    hipLaunchKernelGGL((k_l2_select<true>), dim3((nF + 255) / 256), dim3(256), 0, c->stream, A, (int32_t*)nullptr, c->dSelOff.as<int64_t>(), c->dMappings.as<mm_mapping>(),
                       (const int64_t*)nullptr, 0ll, (unsigned long long*)nullptr, (const unsigned long long*)nullptr);
    MM_HIP(c, hipGetLastError());
  }
  c->nMappings = (size_t)total;
  return MM_OK;
}

extern "C" {

int mm_set_replay_tables(mm_ctx* c, const uint8_t* accept, const int16_t* minIsz, size_t stride) {
  if (!accept || !minIsz || stride != (size_t)c->P.sketchSize + 1) { c->err = "mm_set_replay_tables: need (sketchSize+1)^2 entries per table"; return MM_ERR_ARG; }
  MM_HIP(c, hipSetDevice(c->device));
  MM_HIP(c, c->dAccept.ensure(stride * stride)); MM_HIP(c, c->dMinIsz.ensure(stride * stride * 2));
  MM_HIP(c, hipMemcpyAsync(c->dAccept.p, accept, stride * stride, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemcpyAsync(c->dMinIsz.p, minIsz, stride * stride * 2, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  c->haveReplayTables = true;
  return MM_OK;
}

int mm_mappings_count(const mm_ctx* c, size_t* n) {
  if (!c->mapped || !c->haveReplayTables) return MM_ERR_STATE;
  if (n) *n = c->nMappings;
  return MM_OK;
}

int mm_mappings_download(mm_ctx* c, mm_mapping* out, size_t cap, size_t* n) {
  if (!c->mapped || !c->haveReplayTables) { c->err = "mm_mappings_download: nothing mapped, or mm_set_replay_tables / mm_set_tables_default was not called"; return MM_ERR_STATE; }
  if (n) *n = c->nMappings;
  if (c->nMappings > cap) { c->err = "mm_mappings_download: destination too small"; return MM_ERR_ARG; }
  MM_HIP(c, hipSetDevice(c->device));
  if (c->nMappings) {
    MM_HIP(c, hipMemcpyAsync(out, c->dMappings.p, c->nMappings * sizeof(mm_mapping), hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
  }
  return MM_OK;
}

int mm_mappings_device(const mm_ctx* c, const mm_mapping** dMappings, size_t* n) {
  if (!c->mapped || !c->haveReplayTables) return MM_ERR_STATE;
  if (dMappings) *dMappings = c->dMappings.as<mm_mapping>();
  if (n) *n = c->nMappings;
  return MM_OK;
}

}  // extern "C"

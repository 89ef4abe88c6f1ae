// mashmap_amd/csrc/mm_l2.hip -- L2 stage on the device (gfx950).
//
//   computeL2MappedRegions   src/map/include/computeMap.hpp:1276-1451
//   SlideMapper              src/map/include/slidingMap.hpp:28-212
//
// The reference walks minmerIndex once per L1 candidate, keeps the open reference minmers in a heap ordered
// by wpos_end and, for every record, binary-searches the query sketch twice (insert + eviction).  Here:
//
//   k_l2_extents   thread / candidate : the two record ranges a candidate touches
//                                       (insert stream: minmerIndex order; eviction stream: same records ordered by wpos_end)
//   mm_scan        exclusive scan of the per-candidate op counts
//   k_l2_locate    wave / candidate   : query sketch staged in LDS; every record of both ranges is located in it
//                                       (coalesced hash loads, LDS binary search) and reduced to a 16-bit "op"
//   k_l2_sweep     lane / candidate   : the sequential SlideMapper sweep over the pre-located ops; per-lane state in
//                                       LDS, transposed so that any cell index is bank-conflict free
//
// so the latency-bound pointer chasing of the sweep (two dependent 8-step searches per record) becomes a
// throughput-bound, coalesced pre-pass.
#include "mm_internal.h"
#include "mm_device.h"

#define MM_LOCAP 8          // private L2 locus slots per candidate before the final compaction

struct L2Info { int64_t it0; int64_t itE0; int32_t nIns; int32_t nDel; };
struct L2Tmp { int32_t start, end, shared, strand; };

// op layout: bits 0..10 = 1-based position j of the hash in the query sketch (0: beyond the sketch -> no-op),
//            bit 11 = hash equals q[j], bits 12..13 = query strand + 1
#define OP_J(op) ((int)((op) & 0x7FFu))
#define OP_MATCH(op) ((int)(((op) >> 11) & 1u))
#define OP_QS(op) ((int)(((op) >> 12) & 3u) - 1)

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_l2_extents(int nCand, int segLength, const mm_l1_candidate* __restrict__ l1, const int2* __restrict__ recW,
             const int32_t* __restrict__ recEw, const int64_t* __restrict__ contigOff, L2Info* __restrict__ info, int32_t* __restrict__ cnt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nCand) return;
  const mm_l1_candidate cand = l1[c];
  const int64_t cb = contigOff[cand.seqId], ce = contigOff[cand.seqId + 1];
  // std::lower_bound(minmerIndex, (seqId, rangeStart - segLength - 1))  (computeMap.hpp:1290-1293)
  const int target = cand.rangeStartPos - segLength - 1;
  int64_t lo = cb, hi = ce;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (recW[mid].x < target) lo = mid + 1; else hi = mid; }
  const int64_t it0 = lo;
  hi = ce;                                             // records are visited while wpos <= rangeEnd (:1340)
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (recW[mid].x <= cand.rangeEndPos) lo = mid + 1; else hi = mid; }
  const int64_t itEnd = lo;
  lo = cb; hi = ce;                                    // evictions: wpos_end > rangeStart (anything earlier is never opened, :1325)
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (recEw[mid] <= cand.rangeStartPos) lo = mid + 1; else hi = mid; }
  const int64_t itE0 = lo;
  hi = ce;                                             // ... and wpos_end <= the last visited wpos <= rangeEnd
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (recEw[mid] <= cand.rangeEndPos) lo = mid + 1; else hi = mid; }
  L2Info o; o.it0 = it0; o.itE0 = itE0; o.nIns = (int32_t)(itEnd - it0); o.nDel = (int32_t)(lo - itE0);
  info[c] = o;
  cnt[c] = o.nIns + o.nDel;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan int32 -> int64 (three small kernels; n up to a few hundred million)
// ---------------------------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_TILE (256 * SCAN_ITEMS)
__global__ void __launch_bounds__(256)
k_scan_tiles(int64_t n, const int32_t* __restrict__ in, int64_t* __restrict__ out, int64_t* __restrict__ tileSum) {
  __shared__ int64_t sm[256];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int64_t v[SCAN_ITEMS]; int64_t acc = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = acc; acc += (base + i < n) ? (int64_t)in[base + i] : 0; }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int64_t t = (int)threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  const int64_t excl = sm[threadIdx.x] - acc;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) out[base + i] = excl + v[i];
  if (threadIdx.x == 255) tileSum[blockIdx.x] = sm[255];
}
__global__ void __launch_bounds__(1024)
k_scan_top(int64_t nTiles, int64_t* __restrict__ tileSum, int64_t* __restrict__ total) {
  __shared__ int64_t sm[1024];
  int64_t carry = 0;
  for (int64_t b0 = 0; b0 < nTiles; b0 += 1024) {
    const int64_t i = b0 + threadIdx.x;
    const int64_t x = i < nTiles ? tileSum[i] : 0;
    sm[threadIdx.x] = x;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int64_t t = (int)threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nTiles) tileSum[i] = carry + sm[threadIdx.x] - x;
    const int64_t blockTotal = sm[1023];
    __syncthreads();
    carry += blockTotal;
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(256)
k_scan_add(int64_t n, int64_t* __restrict__ out, const int64_t* __restrict__ tileSum) {
  const int64_t add = tileSum[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) out[base + i] += add;
}

// ---------------------------------------------------------------------------------------------
// k_l2_locate: one wave per candidate (4 per workgroup).  The fragment's query sketch (<= 8 KB) is staged in
// LDS once; the record hashes of both streams are read coalesced and located with an LDS binary search.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_l2_locate(int nCand, int s, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats,
            const uint64_t* __restrict__ qHash, const int8_t* __restrict__ qStrand, const uint64_t* __restrict__ recH,
            const uint64_t* __restrict__ recEh, const L2Info* __restrict__ info, const int64_t* __restrict__ opOff,
            uint16_t* __restrict__ ops) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint64_t* q = (uint64_t*)smem + (size_t)wave * s;
  int8_t* qs = (int8_t*)((uint64_t*)smem + (size_t)4 * s) + (size_t)wave * s;
  for (int c0 = blockIdx.x * 4; c0 < nCand; c0 += gridDim.x * 4) {      // uniform trip count across the workgroup
    const int c = c0 + wave;
    const bool act = c < nCand;
    int S = 0, f = 0; L2Info in{0, 0, 0, 0}; int64_t off = 0;
    if (act) { f = l1[c].frag; S = stats[f].sketchSize; in = info[c]; off = opOff[c]; }
    __syncthreads();                                                      // previous iteration done with q
    for (int p = lane; p < S; p += 64) { q[p] = qHash[(size_t)f * s + p]; qs[p] = qStrand[(size_t)f * s + p]; }
    __syncthreads();
    if (!act) continue;
    const uint64_t qmax = q[S - 1];
    const int total = in.nIns + in.nDel;
    for (int i = lane; i < total; i += 64) {
      const bool isIns = i < in.nIns;
      const uint64_t h = isIns ? recH[in.it0 + i] : recEh[in.itE0 + (i - in.nIns)];
      uint32_t op = 0;
      if (h <= qmax) {
        int lo = 0, hi = S;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid] < h) lo = mid + 1; else hi = mid; }
        op = (uint32_t)(lo + 1) | (q[lo] == h ? 0x800u : 0u) | ((uint32_t)((int)qs[lo] + 1) << 12);
      }
      ops[off + i] = (uint16_t)op;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_l2_sweep: one lane per candidate.  Per-lane SlideMapper state in LDS, cell p of lane l at word p*64+l:
//   bits 0..15 num_before_inc   bit 16 active   bits 24..31 strand_vote (int8)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_l2_sweep(int nCand, int segLength, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats,
           const int2* __restrict__ recW, const int32_t* __restrict__ recEw, const int64_t* __restrict__ contigOff,
           const L2Info* __restrict__ info, const int64_t* __restrict__ opOff, const uint16_t* __restrict__ ops,
           const int64_t* __restrict__ l1Off, L2Tmp* __restrict__ tmp, mm_l2_locus* __restrict__ l2, unsigned long long l2Cap,
           unsigned long long* __restrict__ counters /* [4] l2 cursor, [5] overflow, [6] locus-slot overflow */) {
  extern __shared__ __attribute__((aligned(16))) uint32_t cell[];
  const int lane = threadIdx.x;
  const int cIdx = blockIdx.x * 64 + lane;
  if (cIdx >= nCand) return;
  const mm_l1_candidate cand = l1[cIdx];
  const int f = cand.frag;
  const int S = stats[f].sketchSize;
  const L2Info in = info[cIdx];
  const uint16_t* opS = ops + opOff[cIdx];
  const uint16_t* opE = opS + in.nIns;
  const int2* rw = recW + in.it0;
  const int32_t* ew = recEw + in.itE0;
  const int64_t ce = contigOff[cand.seqId + 1];
#define CELL(p) cell[(p) * 64 + lane]
  CELL(0) = 0;
  for (int p = 1; p <= S; p++) CELL(p) = 1u;
  int pivot = S, pivRank = S, shared = 0, votes = 0;

  auto insert = [&](uint32_t op, int rStrand) {       // slidingMap.hpp:125-165
    const int j = OP_J(op);
    if (j == 0) return;
    uint32_t cw = CELL(j);
    if (OP_MATCH(op)) {
      const int v = (int)(int8_t)(cw >> 24) + OP_QS(op) * rStrand;
      cw = (cw & 0x0000FFFFu) | 0x00010000u | ((uint32_t)(uint8_t)(int8_t)v << 24);
      CELL(j) = cw;
      if (j <= pivot) { shared++; votes += v; }
    } else {
      CELL(j) = cw + 1u;
      if (j <= pivot) pivRank++;
      if (pivRank > S) {
        const uint32_t pw = (pivot == j) ? cw + 1u : CELL(pivot);
        shared -= (int)((pw >> 16) & 1u); votes -= (int)(int8_t)(pw >> 24); pivRank -= (int)(pw & 0xFFFFu); pivot--;
      }
    }
  };
  auto remove = [&](uint32_t op) {                     // slidingMap.hpp:171-211
    const int j = OP_J(op);
    if (j == 0) return;
    const uint32_t cw = CELL(j);
    if (OP_MATCH(op)) {
      if (j <= pivot) { shared--; votes -= (int)(int8_t)(cw >> 24); }
      CELL(j) = cw & 0x0000FFFFu;
    } else {
      CELL(j) = cw - 1u;
      if (j <= pivot) pivRank--;
      if (pivot + 1 <= S) {
        const uint32_t nw = (pivot + 1 == j) ? cw - 1u : CELL(pivot + 1);
        if (pivRank + (int)(nw & 0xFFFFu) <= S) { pivot++; shared += (int)((nw >> 16) & 1u); votes += (int)(int8_t)(nw >> 24); pivRank += (int)(nw & 0xFFFFu); }
      }
    }
  };

  int i = 0, e = 0;
  // pre-load (:1323-1338): records left of the range that are still open at rangeStart
  for (; i < in.nIns; i++) {
    const int2 w = rw[i];
    if (!(w.x < cand.rangeStartPos)) break;
    if ((int)((uint32_t)w.y & 0x7fffffffu) > cand.rangeStartPos) insert(opS[i], w.y < 0 ? -1 : 1);
  }
  // slide (:1340-1434)
  int bestShared = 1; bool inRun = false;
  int curStart = 0, curEnd = 0, curShared = 0;
  int nFlushed = 0; bool havePend = false; L2Tmp pend{0, 0, 0, 0};
  L2Tmp* mySlots = tmp + (size_t)cIdx * MM_LOCAP;
  bool slotOverflow = false;
  auto close_run = [&](int strand) {                   // :1417-1426 / :1440-1449
    if (!havePend || pend.end + segLength < curStart) {
      if (havePend) { if (nFlushed < MM_LOCAP) mySlots[nFlushed] = pend; else slotOverflow = true; nFlushed++; }
      pend.start = curStart; pend.end = curEnd; pend.shared = curShared; pend.strand = strand; havePend = true;
    } else {
      pend.end = curEnd;
    }
  };
  int2 w = i < in.nIns ? rw[i] : make_int2(0, 0);
  for (; i < in.nIns; i++) {
    // next record of the same contig, or this one when it is the contig's last (:1387-1390)
    const int2 wn = (in.it0 + i + 1 < ce) ? rw[i + 1] : w;
    const int prevVotes = votes;
    while (e < in.nDel && ew[e] <= w.x) { remove(opE[e]); e++; }
    insert(opS[i], w.y < 0 ? -1 : 1);
    const int nextW = wn.x;
    if (shared > bestShared) {
      nFlushed = 0; havePend = false;                  // l2_vec_out.clear()
      inRun = true; bestShared = shared; curShared = shared; curStart = w.x; curEnd = nextW;
    } else if (shared == bestShared) {
      if (!inRun) { curShared = shared; curStart = w.x; }
      inRun = true; curEnd = nextW;
    } else {
      if (inRun) { curEnd = nextW; close_run(prevVotes >= 0 ? 1 : -1); curStart = 0; curEnd = 0; curShared = 0; }
      inRun = false;
    }
    w = wn;
  }
  if (inRun) close_run(votes >= 0 ? 1 : -1);
#undef CELL
  const int total = nFlushed + (havePend ? 1 : 0);
  if (slotOverflow) atomicOr(&counters[6], 1ull);
  if (total > 0 && !slotOverflow) {
    const unsigned long long base = atomicAdd(&counters[4], (unsigned long long)total);
    if (base + total > l2Cap) { atomicOr(&counters[5], 1ull); return; }
    const int candLocal = (int)(cIdx - l1Off[f]);
    for (int k = 0; k < total; k++) {
      const L2Tmp t = (k < nFlushed) ? mySlots[k] : pend;
      mm_l2_locus o;
      o.frag = f; o.cand = candLocal; o.seqId = cand.seqId; o.optimalStart = t.start; o.optimalEnd = t.end;
      o.meanOptimalPos = (t.start + t.end) / 2; o.sharedSketchSize = t.shared; o.strand = t.strand;
      l2[base + k] = o;                                // a candidate's loci are contiguous and in emission order
    }
  }
}

// ---------------------------------------------------------------------------------------------
static int scan_i32_to_i64(mm_ctx* c, int64_t n, const int32_t* dIn, int64_t* dOut, int64_t* total) {
  const int64_t nTiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  MM_HIP(c, c->dScanTmp.ensure((size_t)(nTiles + 2) * 8));
  int64_t* tileSum = c->dScanTmp.as<int64_t>();
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)nTiles), dim3(256), 0, c->stream, n, dIn, dOut, tileSum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, nTiles, tileSum, tileSum + nTiles);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nTiles), dim3(256), 0, c->stream, n, dOut, tileSum);
  MM_HIP(c, hipGetLastError());
  MM_HIP(c, hipMemcpyAsync(total, tileSum + nTiles, 8, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  return MM_OK;
}

int mm_launch_l2(mm_ctx* c, unsigned long long* cnt) {
  const DeviceIndex& I = c->idx;
  const int s = c->P.sketchSize;
  const int nC = (int)c->nL1;
  MM_HIP(c, c->dL2Info.ensure((size_t)nC * sizeof(L2Info) + 64));
  MM_HIP(c, c->dL2Cnt.ensure((size_t)nC * 4 + 64));
  MM_HIP(c, c->dL2Off.ensure((size_t)nC * 8 + 64));
  MM_HIP(c, c->dL2Tmp.ensure((size_t)nC * MM_LOCAP * sizeof(L2Tmp) + 64));
  int64_t totalOps = 0;
  {
    KernelTimer t(c, MM_K_L2_LOCATE);
    hipLaunchKernelGGL(k_l2_extents, dim3((nC + 255) / 256), dim3(256), 0, c->stream, nC, c->P.segLength, c->dL1.as<mm_l1_candidate>(),
                       I.recW.as<int2>(), I.recEw.as<int32_t>(), I.contigOff.as<int64_t>(), c->dL2Info.as<L2Info>(), c->dL2Cnt.as<int32_t>());
    MM_HIP(c, hipGetLastError());
    int rc = scan_i32_to_i64(c, nC, c->dL2Cnt.as<int32_t>(), c->dL2Off.as<int64_t>(), &totalOps);
    if (rc != MM_OK) return rc;
    MM_HIP(c, c->dL2Ops.ensure((size_t)totalOps * 2 + 64));
    const size_t ldsLoc = (size_t)4 * s * 9 + 16;
    int blocks = (nC + 3) / 4; if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_l2_locate, dim3(blocks), dim3(256), ldsLoc, c->stream, nC, s, c->dL1.as<mm_l1_candidate>(), c->dStats.as<mm_frag_stats>(),
                       c->dQHash.as<uint64_t>(), c->dQStrand.as<int8_t>(), I.recH.as<uint64_t>(), I.recEh.as<uint64_t>(),
                       c->dL2Info.as<L2Info>(), c->dL2Off.as<int64_t>(), c->dL2Ops.as<uint16_t>());
    MM_HIP(c, hipGetLastError());
  }
  const size_t ldsL2 = (size_t)(s + 1) * 64 * 4;
  if (ldsL2 > 160 * 1024) { c->err = "sketchSize too large for the LDS-resident L2 state"; return MM_ERR_ARG; }
  MM_HIP(c, hipFuncSetAttribute((const void*)k_l2_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsL2));
  if (c->l2Cap < c->nL1 * 2 + 1024) c->l2Cap = c->nL1 * 2 + 1024;
  unsigned long long hc[8];
  for (int attempt = 0; attempt < 8; attempt++) {
    MM_HIP(c, c->dL2.ensure(c->l2Cap * sizeof(mm_l2_locus) + 64));
    MM_HIP(c, hipMemsetAsync(cnt + 4, 0, 24, c->stream));
    {
      KernelTimer t(c, MM_K_L2);
      hipLaunchKernelGGL(k_l2_sweep, dim3((unsigned)((nC + 63) / 64)), dim3(64), ldsL2, c->stream, nC, c->P.segLength, c->dL1.as<mm_l1_candidate>(),
                         c->dStats.as<mm_frag_stats>(), I.recW.as<int2>(), I.recEw.as<int32_t>(), I.contigOff.as<int64_t>(), c->dL2Info.as<L2Info>(),
                         c->dL2Off.as<int64_t>(), c->dL2Ops.as<uint16_t>(), c->dL1Off.as<int64_t>(), c->dL2Tmp.as<L2Tmp>(),
                         c->dL2.as<mm_l2_locus>(), (unsigned long long)c->l2Cap, cnt);
      MM_HIP(c, hipGetLastError());
    }
    MM_HIP(c, hipMemcpyAsync(hc, cnt, 64, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
    if (hc[5]) { c->l2Cap = (size_t)hc[4] + (size_t)hc[4] / 8 + 1024; continue; }
    break;
  }
  if (hc[6]) { c->err = "more than MM_LOCAP tied L2 loci for one candidate"; return MM_ERR_CAPACITY; }
  if (hc[5]) { c->err = "L2 locus buffer overflow"; return MM_ERR_CAPACITY; }
  c->nL2 = (size_t)hc[4];
  return MM_OK;
}

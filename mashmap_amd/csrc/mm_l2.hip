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
//   scan           exclusive scan of the per-candidate entry counts
//   k_l2_locate    wave / candidate   : merges the two streams into ONE time-ordered stream of 8-byte entries.  The query
//                                       sketch and both position arrays are staged in LDS; every record is located in the
//                                       sketch (LDS binary search) and its slot in the merged order is computed by counting
//                                       (eviction e precedes insert i  <=>  wpos_end[e] <= wpos[i], computeMap.hpp:1344-1367)
//   k_l2_sweep     lane / candidate   : the sequential SlideMapper sweep over the merged stream; 64 bytes (8 entries) per lane
//                                       are fetched per step and the next step is prefetched while the current one is
//                                       consumed; per-lane SlideMapper state sits in LDS as 16-bit cells laid out so that a
//                                       lane always hits its own bank
//
// so the latency-bound pointer chasing of the reference (two dependent 8-step searches per record plus a heap) becomes a
// throughput-bound, coalesced pre-pass followed by a sweep whose only memory traffic is one sequential stream per lane.
#include "mm_internal.h"
#include "mm_device.h"

#define MM_LOCAP 8          // private L2 locus slots per candidate before the final compaction

struct L2Info { int64_t it0; int64_t itE0; int32_t nIns; int32_t nDel; };
struct L2Tmp { int32_t start, end, shared, strand; };

// merged-stream entry (uint64):
//   bits 0..10  1-based position j of the hash in the query sketch (0: beyond the sketch -> no effect on the state)
//   bit  11     hash equals q[j]
//   bits 12..13 query strand + 1
//   bits 16..17 type: 0 eviction, 1 insert + evaluate, 2 end of stream, 3 pre-load insert (computeMap.hpp:1323-1338)
//   bit  18     reference strand is REV
//   bits 32..63 wpos of the record (insert), or of the record after the last one (end)
#define E_DEL 0u
#define E_INS 1u
#define E_END 2u
#define E_PRE 3u
#define OP_J(op) ((int)((op) & 0x7FFu))
#define OP_MATCH(op) ((int)(((op) >> 11) & 1u))
#define OP_QS(op) ((int)(((op) >> 12) & 3u) - 1)
#define OP_TYPE(op) (((op) >> 16) & 3u)
#define OP_RSTRAND(op) ((((op) >> 18) & 1u) ? -1 : 1)

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
  // merged stream: every insert, every eviction, the end marker; padded to a whole number of 64-byte sweep steps
  cnt[c] = (o.nIns + o.nDel + 1 + 7) & ~7;
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
// k_l2_locate: one wave per candidate (4 per workgroup).  LDS per wave: the fragment's query sketch (9 B per entry) and,
// when they fit (posCap int32 words), the wpos of the insert range and the wpos_end of the eviction range; candidates with
// longer ranges search those two arrays in global memory instead (same code, X = pointer + stride).
// ---------------------------------------------------------------------------------------------
struct PosArr {            // int32 array with an element stride (LDS copy: stride 1; recW.x in global memory: stride 2)
  const int32_t* p; int stride;
  __device__ __forceinline__ int operator[](int i) const { return p[(size_t)i * stride]; }
};
// #{ i in [0,n) : a[i] < v }   /   #{ i : a[i] <= v }
__device__ __forceinline__ int mm_count_lt(const PosArr a, int n, int v) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ int mm_count_le(const PosArr a, int n, int v) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
  return lo;
}

__global__ void __launch_bounds__(256)
k_l2_locate(int nCand, int s, int posCap, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats,
            const uint64_t* __restrict__ qHash, const int8_t* __restrict__ qStrand, const uint64_t* __restrict__ recH,
            const int2* __restrict__ recW, const uint64_t* __restrict__ recEh, const int32_t* __restrict__ recEw,
            const int64_t* __restrict__ contigOff, const L2Info* __restrict__ info, const int64_t* __restrict__ opOff,
            uint64_t* __restrict__ ops) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // per-wave carve: q[s] u64 | pos[posCap] i32 | qs[s] i8
  const size_t perWave = (size_t)s * 8 + (size_t)posCap * 4 + (((size_t)s + 15) & ~(size_t)15);
  unsigned char* base = smem + (size_t)wave * perWave;
  uint64_t* q = (uint64_t*)base;
  int32_t* posL = (int32_t*)(base + (size_t)s * 8);
  int8_t* qs = (int8_t*)(base + (size_t)s * 8 + (size_t)posCap * 4);
  for (int c = blockIdx.x * 4 + wave; c < nCand; c += gridDim.x * 4) {    // waves are independent: no workgroup barrier below
    const mm_l1_candidate cand = l1[c];
    const int f = cand.frag;
    const int S = stats[f].sketchSize;
    const L2Info in = info[c];
    uint64_t* out = ops + opOff[c];
    __threadfence_block();                                                 // previous candidate's LDS reads are done
    for (int p = lane; p < S; p += 64) { q[p] = qHash[(size_t)f * s + p]; qs[p] = qStrand[(size_t)f * s + p]; }
    const bool inLds = in.nIns + in.nDel <= posCap;
    PosArr insX, delX;
    if (inLds) {
      for (int i = lane; i < in.nIns; i += 64) posL[i] = recW[in.it0 + i].x;
      for (int e = lane; e < in.nDel; e += 64) posL[in.nIns + e] = recEw[in.itE0 + e];
      insX.p = posL; insX.stride = 1; delX.p = posL + in.nIns; delX.stride = 1;
    } else {
      insX.p = (const int32_t*)(recW + in.it0); insX.stride = 2; delX.p = recEw + in.itE0; delX.stride = 1;
    }
    __threadfence_block();
    const uint64_t qmax = q[S - 1];
    auto locate = [&](uint64_t h) -> uint32_t {
      if (h > qmax) return 0u;
      int lo = 0, hi = S;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid] < h) lo = mid + 1; else hi = mid; }
      return (uint32_t)(lo + 1) | (q[lo] == h ? 0x800u : 0u) | ((uint32_t)((int)qs[lo] + 1) << 12);
    };
    // records left of the range (wpos < rangeStart): only those still open at rangeStart are pre-loaded (:1323-1338)
    const int nPreAll = mm_count_lt(insX, in.nIns, cand.rangeStartPos);
    int nPre = 0;
    for (int b0 = 0; b0 < nPreAll; b0 += 64) {
      const int i = b0 + lane;
      bool keep = false; int2 w = make_int2(0, 0);
      if (i < nPreAll) { w = recW[in.it0 + i]; keep = (int)((uint32_t)w.y & 0x7fffffffu) > cand.rangeStartPos; }
      const uint64_t m = __ballot(keep);
      if (keep) {
        const uint32_t op = locate(recH[in.it0 + i]) | (E_PRE << 16) | (w.y < 0 ? (1u << 18) : 0u);
        out[nPre + (int)mm_popc_below(m)] = ((uint64_t)(uint32_t)w.x << 32) | op;
      }
      nPre += __popcll(m);
    }
    const int nMain = in.nIns - nPreAll;
    PosArr mainX = insX; mainX.p += (size_t)nPreAll * insX.stride;
    for (int mi = lane; mi < nMain; mi += 64) {                            // inserts that are evaluated
      const int2 w = recW[in.it0 + nPreAll + mi];
      const uint32_t op = locate(recH[in.it0 + nPreAll + mi]) | (E_INS << 16) | (w.y < 0 ? (1u << 18) : 0u);
      out[nPre + mi + mm_count_le(delX, in.nDel, w.x)] = ((uint64_t)(uint32_t)w.x << 32) | op;
    }
    for (int e = lane; e < in.nDel; e += 64) {                             // evictions that happen before some insert
      const int ew = delX[e];
      const int before = mm_count_lt(mainX, nMain, ew);                    // inserts with wpos < wpos_end[e] come first
      if (before < nMain) out[nPre + e + before] = (uint64_t)(locate(recEh[in.itE0 + e]) | (E_DEL << 16));
    }
    if (lane == 0) {                                                       // end marker right behind the last insert
      int endPos = nPre, nextW = 0;
      if (nMain > 0) {
        const int wl = mainX[nMain - 1];
        endPos = nPre + nMain + mm_count_le(delX, in.nDel, wl);
        // wpos of the next record of the same contig, or of the last one when it closes the contig (:1387-1390)
        nextW = (in.it0 + in.nIns < contigOff[cand.seqId + 1]) ? recW[in.it0 + in.nIns].x : wl;
      }
      out[endPos] = ((uint64_t)(uint32_t)nextW << 32) | (uint64_t)(E_END << 16);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_l2_sweep: one lane per candidate.  Per-lane SlideMapper state in LDS, one 16-bit cell per query position p:
//   bits 0..11 num_before_inc   bit 12 active   bits 13..14 strand_vote + 1
// cell p of lane l lives in dword p*32 + (l & 31), half l >> 5: both 32-lane halves of a DS instruction see 32 distinct banks.
// ---------------------------------------------------------------------------------------------
#define CELL_CNT(x) ((int)((x) & 0xFFFu))
#define CELL_ACT(x) ((int)(((x) >> 12) & 1u))
#define CELL_VOTE(x) ((int)(((x) >> 13) & 3u) - 1)

__global__ void __launch_bounds__(64)
k_l2_sweep(int nCand, int segLength, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats,
           const int64_t* __restrict__ opOff, const int32_t* __restrict__ opCnt, const uint64_t* __restrict__ ops,
           const int64_t* __restrict__ l1Off, L2Tmp* __restrict__ tmp, mm_l2_locus* __restrict__ l2, unsigned long long l2Cap,
           unsigned long long* __restrict__ counters /* [4] l2 cursor, [5] overflow, [6] locus-slot overflow */) {
  extern __shared__ __attribute__((aligned(16))) uint16_t cell[];
  const int lane = threadIdx.x;
  const int cIdx = blockIdx.x * 64 + lane;
  if (cIdx >= nCand) return;
  const mm_l1_candidate cand = l1[cIdx];
  const int f = cand.frag;
  const int S = stats[f].sketchSize;
  const uint4* src = (const uint4*)(ops + opOff[cIdx]);
  const int nSteps = opCnt[cIdx] >> 3;                 // 8 entries = 4 x 16 bytes per step
  const int lbase = (lane & 31) * 2 + (lane >> 5);
#define CELL(p) cell[(p) * 64 + lbase]
  CELL(0) = 0;
  for (int p = 1; p <= S; p++) CELL(p) = 1u | (1u << 13);       // num_before_inc = 1, inactive, vote 0
  int pivot = S, pivRank = S, shared = 0, votes = 0;
  bool doubleOpen = false;

  // SlideMapper::insert_minmer / delete_minmer (slidingMap.hpp:125-211) as straight-line code: the four cases (insert or
  // delete x hash matches a query hash or not) are selected arithmetically, because the 64 lanes of a wave are at different
  // candidates and take different cases at every entry -- as branches they serialise (measured 5x slower).  The cell of the
  // hash, of the pivot and of its right neighbour are read up front, independent of the case.
  auto apply = [&](uint32_t op, bool isIns) {
    const int j = OP_J(op);                            // 0: hash beyond the query sketch -> no effect (cell 0 is a dummy)
    const bool valid = j != 0, match = OP_MATCH(op) != 0;
    const int pn = pivot + 1 <= S ? pivot + 1 : S;
    const uint32_t cw = CELL(j), pw = CELL(pivot), nw = CELL(pn);
    const int v = OP_QS(op) * OP_RSTRAND(op);          // a query hash has one open reference window at a time (windowLen == 0);
    const bool IM = valid & isIns & match, IN = valid & isIns & !match, DM = valid & !isIns & match, DN = valid & !isIns & !match;
    doubleOpen |= IM & (CELL_ACT(cw) != 0);            // the 2-bit vote relies on it, so a violation is reported, not absorbed
    uint32_t ncw = cw;
    ncw = IM ? ((cw & 0xFFFu) | (1u << 12) | ((uint32_t)(v + 1) << 13)) : ncw;
    ncw = IN ? cw + 1u : ncw;
    ncw = DM ? ((cw & 0xFFFu) | (1u << 13)) : ncw;
    ncw = DN ? cw - 1u : ncw;
    CELL(j) = (uint16_t)ncw;
    const int ip = j <= pivot ? 1 : 0;
    shared += (IM ? ip : 0) - (DM ? ip : 0);
    votes += ((IM & (ip != 0)) ? v : 0) - ((DM & (ip != 0)) ? CELL_VOTE(cw) : 0);
    pivRank += (IN ? ip : 0) - (DN ? ip : 0);
    // insert of a non-shared hash can push the pivot one cell left (:155-160)
    const uint32_t pwp = (pivot == j) ? ncw : pw;
    const bool left = IN & (pivRank > S);
    // delete of a non-shared hash can let it move one cell right (:201-207)
    const uint32_t nwp = (pn == j) ? ncw : nw;
    const bool right = DN & (pivot + 1 <= S) & (pivRank + CELL_CNT(nwp) <= S);
    shared += (right ? CELL_ACT(nwp) : 0) - (left ? CELL_ACT(pwp) : 0);
    votes += (right ? CELL_VOTE(nwp) : 0) - (left ? CELL_VOTE(pwp) : 0);
    pivRank += (right ? CELL_CNT(nwp) : 0) - (left ? CELL_CNT(pwp) : 0);
    pivot += (right ? 1 : 0) - (left ? 1 : 0);
  };

  // best-position bookkeeping (:1376-1449)
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
  // the evaluation of insert i needs the wpos of record i+1, so it is carried until the next insert / end entry arrives
  // lastVotes = strand_votes right after the most recent insert (pre-load included): the reference samples it before the
  // evictions that precede the next insert (:1342)
  bool evalPending = false; int evW = 0, evShared = 0, evPrevVotes = 0, lastVotes = 0;
  auto evaluate = [&](int nextW) {
    if (evShared > bestShared) {
      nFlushed = 0; havePend = false;                  // l2_vec_out.clear()
      inRun = true; bestShared = evShared; curShared = evShared; curStart = evW; curEnd = nextW;
    } else if (evShared == bestShared) {
      if (!inRun) { curShared = evShared; curStart = evW; }
      inRun = true; curEnd = nextW;
    } else {
      if (inRun) { curEnd = nextW; close_run(evPrevVotes >= 0 ? 1 : -1); curStart = 0; curEnd = 0; curShared = 0; }
      inRun = false;
    }
  };

  bool done = false;
  uint4 cur[4], nxt[4];
  if (nSteps > 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = src[k];
  }
  for (int step = 0; step < nSteps && !done; step++) {
    if (step + 1 < nSteps) {
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = src[(size_t)(step + 1) * 4 + k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t lo = (k & 1) ? cur[k >> 1].z : cur[k >> 1].x;
      const int wpos = (int)((k & 1) ? cur[k >> 1].w : cur[k >> 1].y);
      if (done) continue;
      const uint32_t type = OP_TYPE(lo);
      if (type == E_INS || type == E_END) {
        if (evalPending) { evaluate(wpos); evalPending = false; }
        if (type == E_END) { done = true; continue; }
      }
      evPrevVotes = (type == E_INS) ? lastVotes : evPrevVotes;
      apply(lo, type != E_DEL);
      lastVotes = (type != E_DEL) ? votes : lastVotes;
      if (type == E_INS) { evW = wpos; evShared = shared; evalPending = true; }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nxt[k];
  }
  if (inRun) close_run(votes >= 0 ? 1 : -1);
#undef CELL
  int total = nFlushed + (havePend ? 1 : 0);
  if (slotOverflow) { atomicOr(&counters[6], 1ull); total = 0; }
  if (doubleOpen) atomicOr(&counters[6], 2ull);
  // one reservation per wave (64 candidates): exclusive scan of the lanes' counts, lane 63 of the active lanes asks
  int incl = total;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
  const uint64_t activeMask = __ballot(1);
  const int lastLane = 63 - __builtin_clzll(activeMask);
  const int waveTotal = __shfl(incl, lastLane);
  unsigned long long wbase = 0;
  if (lane == lastLane && waveTotal > 0) wbase = atomicAdd(&counters[4], (unsigned long long)waveTotal);
  wbase = ((unsigned long long)(uint32_t)__shfl((int)(wbase >> 32), lastLane) << 32) | (uint32_t)__shfl((int)(uint32_t)wbase, lastLane);
  if (waveTotal > 0 && wbase + (unsigned long long)waveTotal > l2Cap) { if (lane == lastLane) atomicOr(&counters[5], 1ull); return; }
  if (total > 0) {
    const unsigned long long base = wbase + (unsigned long long)(incl - total);
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
int mm_scan_i32_to_i64(mm_ctx* c, int64_t n, const int32_t* dIn, int64_t* dOut, int64_t* total) {
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
    int rc = mm_scan_i32_to_i64(c, nC, c->dL2Cnt.as<int32_t>(), c->dL2Off.as<int64_t>(), &totalOps);
    if (rc != MM_OK) return rc;
    MM_HIP(c, c->dL2Ops.ensure((size_t)totalOps * 8 + 256));
    // LDS for the two position arrays of a candidate: ~4x the sketch size covers the typical candidate (2 records per
    // sketch entry per segLength of range, both streams); longer ones search global memory
    int posCap = 16 * s; if (posCap < 2048) posCap = 2048; if (posCap > 8192) posCap = 8192;
    const size_t perWave = (size_t)s * 8 + (size_t)posCap * 4 + (((size_t)s + 15) & ~(size_t)15);
    const size_t ldsLoc = perWave * 4;
    MM_HIP(c, hipFuncSetAttribute((const void*)k_l2_locate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsLoc));
    int blocks = (nC + 3) / 4; if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_l2_locate, dim3(blocks), dim3(256), ldsLoc, c->stream, nC, s, posCap, c->dL1.as<mm_l1_candidate>(),
                       c->dStats.as<mm_frag_stats>(), c->dQHash.as<uint64_t>(), c->dQStrand.as<int8_t>(), I.recH.as<uint64_t>(),
                       I.recW.as<int2>(), I.recEh.as<uint64_t>(), I.recEw.as<int32_t>(), I.contigOff.as<int64_t>(),
                       c->dL2Info.as<L2Info>(), c->dL2Off.as<int64_t>(), c->dL2Ops.as<uint64_t>());
    MM_HIP(c, hipGetLastError());
  }
  const size_t ldsL2 = (size_t)(s + 1) * 64 * 2;          // cells 0..S
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
                         c->dStats.as<mm_frag_stats>(), c->dL2Off.as<int64_t>(), c->dL2Cnt.as<int32_t>(), c->dL2Ops.as<uint64_t>(),
                         c->dL1Off.as<int64_t>(), c->dL2Tmp.as<L2Tmp>(), c->dL2.as<mm_l2_locus>(), (unsigned long long)c->l2Cap, cnt);
      MM_HIP(c, hipGetLastError());
    }
    MM_HIP(c, hipMemcpyAsync(hc, cnt, 64, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
    if (hc[5]) { c->l2Cap = (size_t)hc[4] + (size_t)hc[4] / 8 + 1024; continue; }
    break;
  }
  if (hc[6] & 2ull) { c->err = "a query hash had two open reference windows at once (index intervals of one hash overlap)"; return MM_ERR_STATE; }
  if (hc[6]) { c->err = "more than MM_LOCAP tied L2 loci for one candidate"; return MM_ERR_CAPACITY; }
  if (hc[5]) { c->err = "L2 locus buffer overflow"; return MM_ERR_CAPACITY; }
  c->nL2 = (size_t)hc[4];
  return MM_OK;
}

// mashmap_amd/csrc/mm_l2.hip -- L2 stage on the device (gfx950).
//
//   computeL2MappedRegions   src/map/include/computeMap.hpp:1276-1451
//   SlideMapper              src/map/include/slidingMap.hpp:28-212
//
// The reference walks minmerIndex once per L1 candidate, keeps the open reference minmers in a heap ordered
// by wpos_end and, for every record, binary-searches the query sketch twice (insert + eviction).  Here the index already
// holds, per contig, the merged stream of insert and eviction events (mm_build_device_index), so a candidate is a slice of it:
//
//   k_l2_extents   thread / candidate : the slice [e0, eMid, ub) of the event stream (three searches, each bracketed to one block of
//                                       positions by evBlock) and the block's list of records open at its start
//   scan           exclusive scan of the per-candidate entry counts
//   k_l2_locate    wave / candidate   : streams the open-record list and the slice (coalesced 16 B per event), locates every hash in
//                                       the query sketch (LDS bucket table + short scan) and writes a compact 4-byte entry per event
//                                       that can matter: pre-load inserts still open at rangeStart, inserts, evictions of hashes inside
//                                       the sketch's range; positions are delta-coded against the previous insert
//   k_l2_sweep     lane / candidate   : the sequential SlideMapper sweep over that stream; 64 bytes (16 entries) per lane
//                                       are fetched per step and the next step is prefetched while the current one is
//                                       consumed; per-lane SlideMapper state sits in LDS as 8-bit cells (16-bit for the rare
//                                       candidate that overflows them) laid out so that a lane always hits its own bank; the state
//                                       update runs on integer lane masks, without branches
//
// so the latency-bound pointer chasing of the reference (two dependent 8-step searches per record plus a heap) becomes a
// streaming pre-pass followed by a sweep whose only memory traffic is one sequential stream per lane.
#include "mm_internal.h"
#include "mm_device.h"
#include "mm_heap.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define MM_LOCAP0 8         // private L2 locus slots per candidate before the final compaction (doubled and re-run on overflow)

struct L2Info { int64_t e0; int32_t nPre; int32_t nAll;         // slice [e0, e0+nAll) of the contig's events, the first nPre before rangeStart
                int32_t sketch;                                 // the fragment's sketchSize, bit 31: no seed was removed (take the raw sketch)
                int32_t nOpen; int64_t open0;                   // records open at the block boundary before rangeStart: op*[open0, open0+nOpen)
                int32_t target; int32_t pad; };                 // pre-load takes records with wpos >= target = rangeStart - segLength - 1 (:1290)
struct L2Tmp { int32_t start, end, shared, strand; };
// The SlideMapper state of a candidate once the pre-load (computeMap.hpp:1323-1338: the records still open at rangeStart) is in, in closed
// form.  With inserts only the state does not depend on their order -- cell counts add up, the pivot is the largest p with
// p + #{reference-only hashes below q[p]} <= S (slidingMap.hpp:155-160 keeps exactly that), shared / votes are sums over the active cells
// up to it -- so k_l2_locate, which has the candidate's sketch in LDS anyway, builds it in parallel instead of writing ~s pre-load entries
// that the sweep then applies one by one (a fifth to a third of a stream).  Cells: 16 bit each, count (12) | active << 12 | (vote + 1) << 13
// = the wide sweep's cell format; row of s + 2 cells per candidate of a chunk.  flags: 1 = a query hash was open twice in the pre-load (the
// 2-bit vote cannot hold it: k_l2_sweep_exact replays the pre-load from the index), 2 = a pre-load of 4000 records or more,
// which the 12-bit counts might not hold (likewise).
struct L2Init { int32_t pivot, pivRank, shared, votes; };
#define L2INIT_FLAG_SHIFT 24        // flags travel in the top byte of `pivot` (pivot <= 8190)

// stream entry (uint32), JB = width of the sketch-position field (11 for sketches of up to 2046 entries, 13 beyond):
//   bits 0..JB-1        1-based position j of the hash in the query sketch (0: beyond the sketch -> no effect on the state)
//   bit  JB             hash equals q[j]
//   bits JB+1..JB+2     strand vote of a matching insert + 1 (query strand x reference strand)
//   bits JB+3..26       insert / end: wpos minus the running position (previous insert's wpos, rangeStart at first)
//   bits 27..31         kind, one bit each (the sweep turns a bit into a lane mask with one v_bfe_i32):
//                       insert + evaluate, eviction, pre-load insert (computeMap.hpp:1323-1338), end of stream, skip
//   skip: adds its low 27 bits to the running position (a gap that does not fit the delta field)
#define E_INS_BIT 27
#define E_DEL_BIT 28
#define E_PRE_BIT 29
#define E_END_BIT 30
#define E_SKIP_BIT 31
#define E_SKIP_MAX ((1u << 27) - 1u)
#define E_STEP 16           // entries per 64-byte sweep step
template <int JB> struct EF {
  static constexpr uint32_t JMASK = (1u << JB) - 1u;
  static constexpr int MATCH_BIT = JB, VOTE_SHIFT = JB + 1, DELTA_SHIFT = JB + 3, DELTA_BITS = 27 - (JB + 3);
  static constexpr uint32_t MAXDELTA = (1u << DELTA_BITS) - 1u;
};
static inline int mm_l2_jb(int s) { return s > 2046 ? 13 : 11; }

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_l2_extents(int nCand, int segLength, int deltaBits, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats, const uint32_t* __restrict__ evKey,
             const int64_t* __restrict__ contigBlock, const int64_t* __restrict__ blockOff,
             const int64_t* __restrict__ evBlock, L2Info* __restrict__ info, int32_t* __restrict__ cnt, const unsigned long long* __restrict__ nDev,
             unsigned long long* __restrict__ counters /* [6] |= 32: more candidates than the buffers of this (steady-state) pass hold */) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (nDev) {                                                     // steady state: the launch covers the buffers' capacity, the count is on the device
    const long long n = (long long)*nDev;
    if (c == 0 && n > nCand) atomicOr(&counters[6], 32ull);
    if (c >= n) { if (c < nCand) { cnt[c] = 0; info[c].e0 = 0; } return; }
  }
  if (c >= nCand) return;
  const mm_l1_candidate cand = l1[c];
  const mm_frag_stats fst = stats[cand.frag];
  auto lowerIn = [&](int64_t lo, int64_t hi, uint32_t key) { while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (evKey[mid] < key) lo = mid + 1; else hi = mid; } return lo; };
  // std::lower_bound(minmerIndex, (seqId, rangeStart - segLength - 1))  (computeMap.hpp:1290-1293): inserts with wpos >= target are
  // pre-loaded if still open at rangeStart.  Those that start before the block boundary B <= rangeStart come from the block's
  // list of open records, the rest from the events in [max(B, target), rangeStart).  evBlock brackets every search to one block.
  const int target = cand.rangeStartPos - segLength - 1;
  const int64_t blk0 = contigBlock[cand.seqId], nBlk = contigBlock[cand.seqId + 1] - blk0;
  const int64_t blk = blk0 + (cand.rangeStartPos >> MM_OPEN_BLOCK_SHIFT);
  const int64_t ob = blockOff[blk], oe = blockOff[blk + 1];
  const int64_t evB = evBlock[blk], evN = evBlock[blk + 1];
  const int B = (cand.rangeStartPos >> MM_OPEN_BLOCK_SHIFT) << MM_OPEN_BLOCK_SHIFT;
  const int64_t e0 = target > B ? lowerIn(evB, evN, (uint32_t)target * 2u) : evB;
  const int64_t eMid = lowerIn(e0, evN, (uint32_t)cand.rangeStartPos * 2u + 1u);   // first event of the slide: insert at rangeStart or anything later
  int64_t bE = cand.rangeEndPos >> MM_OPEN_BLOCK_SHIFT; if (bE > nBlk - 1) bE = nBlk - 1;
  const int64_t evE = evBlock[blk0 + bE], evEn = evBlock[blk0 + bE + 1];
  const int64_t ub = lowerIn(evE > eMid ? evE : eMid, evEn, (uint32_t)cand.rangeEndPos * 2u + 2u);   // records are visited while wpos <= rangeEnd (:1340)
  L2Info o; o.e0 = e0; o.nPre = (int32_t)(eMid - e0); o.nAll = (int32_t)(ub - e0);
  o.open0 = ob; o.nOpen = (int32_t)(oe - ob); o.target = target;
  o.sketch = fst.sketchSize | (fst.rawSketchSize == fst.sketchSize ? (int32_t)0x80000000 : 0); o.pad = 0;
  info[c] = o;
  // upper bound of the stream: every event, the end marker, one skip per 2^deltaBits of range, slack for the end marker's own skips
  // (the record behind the last insert may be anywhere in the contig: up to 2^31 / 2^27 of them)
  // (the pre-load is not part of the stream: it reaches the sweeps as a ready-made state, L2Init)
  const int n = (o.nAll - o.nPre) + 1 + ((cand.rangeEndPos - cand.rangeStartPos) >> deltaBits) + 20;
  cnt[c] = (n + E_STEP - 1) & ~(E_STEP - 1);
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

// steady-state passes: the streams' total (the scan's last word) against the buffer as the previous pass left it; a batch that does not
// fit is flagged and its candidate count zeroed, so that the kernels behind this one do nothing (the pass is then redone with the
// host's sizing)
__global__ void k_l2_gate(const int64_t* __restrict__ total, int64_t opsCap, unsigned long long* __restrict__ counters /* [2] candidates, [6] |= 8 */) {
  if (*total > opsCap) { counters[6] |= 8ull; counters[2] = 0ull; }
}

// ---------------------------------------------------------------------------------------------
// Sort keys that put the candidates of a chunk in REFERENCE order (the event index where a candidate's slice starts, coarsened).
// k_l2_locate streams ~3 segment lengths of the index per candidate and the reads cover the reference several times over, in random
// order: taken as they come, every slice is fetched from HBM (16 B per event: the kernel sits on the HBM roofline against a human-scale
// index); taken in reference order, the waves in flight at any moment read one neighbourhood of the index, which the Infinity Cache holds.
__global__ void __launch_bounds__(256)
k_l2_pos_keys(int c0, int n, int shift, const L2Info* __restrict__ info, uint32_t* __restrict__ kOut, int32_t* __restrict__ vOut, const unsigned long long* __restrict__ nDev) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool real = !nDev || c0 + i < (int)*nDev;                 // steady state: n is the buffer's capacity, the entries behind the real ones sort last
  kOut[i] = real ? (uint32_t)(info[c0 + i].e0 >> shift) : 0xFFFFu;
  vOut[i] = c0 + i;
}

// ---------------------------------------------------------------------------------------------
// k_l2_locate: one wave per candidate (4 per workgroup, no workgroup barrier).  LDS per wave: the fragment's query sketch
// (hash + strand) and a table of NB >= s equal-width buckets over [0, qmax] that turns the lower_bound of a hash into one table read
// plus a walk of ~1 entry (the hashes of a sketch are uniform, so equal-width buckets are balanced).
// LDS of one wave of k_l2_locate: sketch + sentinel (8 B each), their high words (4 B), NB + 1 bucket starts (2 B), strands (1 B)
__host__ __device__ static inline size_t mm_locate_lds_per_wave(int s, int NB) {
  const size_t b = (size_t)(s + 1) * 8 + (size_t)(s + 2) / 2 * 8 + (size_t)(NB + 4) * 2 + (((size_t)s + 15) & ~(size_t)15) + (((size_t)(s + 4) / 2 * 4 + 15) & ~(size_t)15);
  return (b + 15) & ~(size_t)15;
}
// ---------------------------------------------------------------------------------------------
template <int JB>
__global__ void __launch_bounds__(256)
k_l2_locate(int cBase, int nCand, int64_t opsBase, int s, int NB, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats,
            const uint64_t* __restrict__ qHash, const int8_t* __restrict__ qStrand,
            const uint64_t* __restrict__ skHash, const int8_t* __restrict__ skStrand,
            const uint32_t* __restrict__ evKey, const uint32_t* __restrict__ evAux, const uint64_t* __restrict__ evHash,
            const uint32_t* __restrict__ opKey, const uint32_t* __restrict__ opAux, const uint64_t* __restrict__ opHash,
            const int64_t* __restrict__ contigOff, const L2Info* __restrict__ info, const int64_t* __restrict__ opOff,
            const int32_t* __restrict__ opCnt, uint32_t* __restrict__ ops, const int32_t* __restrict__ order /* candidates in reference order, or null */,
            unsigned long long* __restrict__ counters /* [6] |= 4: a stream outgrew its reservation */,
            const unsigned long long* __restrict__ nDev, uint16_t* __restrict__ initCells, L2Init* __restrict__ initState, int initStride, int preLimit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // (readfirstlane: the candidate's extents then live in scalar registers)
  unsigned char* base = smem + (size_t)wave * mm_locate_lds_per_wave(s, NB);
  uint64_t* q = (uint64_t*)base;                                   // the query sketch + one sentinel
  uint32_t* qhi = (uint32_t*)(base + (size_t)(s + 1) * 8);         // its high words (the bucket walk compares these)
  uint16_t* bkt = (uint16_t*)(base + (size_t)(s + 1) * 8 + (size_t)(s + 2) / 2 * 8);   // bkt[b] = #query hashes whose bucket is < b, b = 0..NB
  int8_t* qs = (int8_t*)(base + (size_t)(s + 1) * 8 + (size_t)(s + 2) / 2 * 8 + (size_t)(NB + 4) * 2);
  uint32_t* ic = (uint32_t*)(base + (size_t)(s + 1) * 8 + (size_t)(s + 2) / 2 * 8 + (size_t)(NB + 4) * 2 + (((size_t)s + 15) & ~(size_t)15));   // pre-load state: two 16-bit cells per dword
  const int wpb = (int)(blockDim.x >> 6);                          // waves per workgroup: 4, fewer when a sketch's LDS share is large
  if (nDev) { const int nd = (int)*nDev - cBase; if (nd < nCand) nCand = nd; }   // steady state: the count stayed on the device (k_l2_gate has zeroed it if the streams do not fit)
  for (int ci = blockIdx.x * wpb + wave; ci < nCand; ci += gridDim.x * wpb) {
    const int c = order ? order[ci] : cBase + ci;                  // this launch covers the candidates [cBase, cBase + nCand): their streams start at ops[opOff - opsBase]
    const mm_l1_candidate cand = l1[c];
    const int f = cand.frag;
    const L2Info in = info[c];
    const int S = in.sketch & 0x7fffffff;
    uint32_t* out = ops + (opOff[c] - opsBase);
    const int cap = opCnt[c];
    const int nOpen = in.nOpen, nPre = in.nPre, nAll = in.nAll;
    // The kernel is bound by the latency of its global reads, not by their volume (the slices come from the Infinity Cache) nor by
    // instruction issue: round 3 walked a chain of ~20 dependent round trips per candidate (sketch, tail scan, the record behind the
    // tail, then one per 64 events).  Everything that depends only on the candidate's extents is therefore requested up front -- the
    // last 64 events of the slice, the 64 behind it, the first chunk of the stream -- and every later chunk while its predecessor
    // is being processed.
    const int64_t ce = contigOff[cand.seqId + 1];
    const int iT = nAll - 64 + lane;
    const uint32_t tailKey = (iT >= nPre && iT < nAll) ? evKey[in.e0 + iT] : 0u;           // insert flag = bit 0; 0 outside the slide
    const uint32_t behindKey = in.e0 + nAll + lane < ce ? evKey[in.e0 + nAll + lane] : 0u;
    // the stream in chunks of 64: the block's open records (all inserts, all before rangeStart), then the events [0, nPre) before
    // rangeStart, then the slide [nPre, nEv).  A chunk reads from wave-uniform arrays (one loop over a concatenation cost a per-lane
    // choice between two sets of three 64-bit pointers in every chunk)
    const int nChO = (nOpen + 63) >> 6, nChPre = nChO + ((nPre + 63) >> 6);
    auto loadChunk = [&](int ch, uint32_t& key, uint32_t& aux, uint64_t& h) {
      key = 0; aux = 0; h = 0;                                       // a lane without a record: no insert flag, nothing kept
      if (ch < nChO) { const int i = ch * 64 + lane; if (i < nOpen) { key = opKey[in.open0 + i]; aux = opAux[in.open0 + i]; h = opHash[in.open0 + i]; } }
      else {
        const int i = ch < nChPre ? (ch - nChO) * 64 + lane : nPre + (ch - nChPre) * 64 + lane;
        if (i < (ch < nChPre ? nPre : nAll)) { key = evKey[in.e0 + i]; aux = evAux[in.e0 + i]; h = evHash[in.e0 + i]; }
      }
    };
    uint32_t nKey, nAux; uint64_t nHash;
    loadChunk(0, nKey, nAux, nHash);
    __threadfence_block();                                         // previous candidate's LDS reads are done
    // a fragment that lost no frequent seed has no copy in qHash/qStrand: its sketch is the raw one (k_lookup_l1)
    const bool raw = in.sketch < 0;
    const uint64_t* srcH = (raw ? skHash : qHash) + (size_t)f * s;
    const int8_t* srcS = (raw ? skStrand : qStrand) + (size_t)f * s;
    for (int p = lane; p < S; p += 64) { const uint64_t h = srcH[p]; q[p] = h; qhi[p] = (uint32_t)(h >> 32); qs[p] = srcS[p]; }
    if (lane == 0) { q[S] = ~0ull; qhi[S] = ~0u; }                 // sentinel: a walk for h <= qmax needs no end test
    for (int d = lane; d <= (S + 1) / 2; d += 64) {                // cells 0 and S + 1: empty; 1 .. S: count 1, inactive, vote 0 (SlideMapper::init, slidingMap.hpp:103-121)
      const int p0 = 2 * d, p1 = 2 * d + 1;
      ic[d] = ((p0 >= 1 && p0 <= S) ? 0x2001u : 0u) | (((p1 >= 1 && p1 <= S) ? 0x2001u : 0u) << 16);
    }
    __threadfence_block();
    const uint64_t qmax = q[S - 1];
    // bucket(h): monotone map of [0, qmax] onto 0..NB-1: the top 24 significant bits times M >> 32, M <= 2^32 * NB / (top24(qmax) + 1)
    // (any smaller M stays monotone and below NB; the float estimate is shaded down)
    const int sh = qmax ? (int)__builtin_clzll(qmax) : 63;
    // (NB <= 16384 and the top bit of top24(qmax) is set, so M < 2^24 as well: a 24 x 24-bit multiply, which issues at full rate)
    const uint32_t bM = (uint32_t)((float)NB * 4294967296.0f / ((float)(uint32_t)((qmax << sh) >> 40) + 1.0f) * 0.99999f) & 0xFFFFFFu;
    auto bucket = [&](uint64_t h) -> int { return (int)(((uint64_t)((uint32_t)((h << sh) >> 40) & 0xFFFFFFu) * (uint64_t)bM) >> 32); };
    // bkt[b] = #{p : bucket(q[p]) < b}.  q is sorted, so entry p owns the buckets (bucket(q[p-1]), bucket(q[p])] and the
    // sentinel p = S owns the rest up to NB: a scatter of ~NB/S stores per lane instead of NB + 1 binary searches
    for (int p0 = 0; p0 <= S; p0 += 64) {
      const int p = p0 + lane;
      if (p <= S) {
        const int from = p == 0 ? 0 : bucket(q[p - 1]) + 1;
        const int to = p == S ? NB : bucket(q[p]);
        for (int b = from; b <= to; b++) bkt[b] = (uint16_t)p;
      }
    }
    __threadfence_block();
    auto locate = [&](uint64_t h) -> uint32_t {
      if (h > qmax) return 0u;
      const int b = bucket(h);
      // lower_bound(q, h): everything in earlier buckets is smaller and h <= qmax < sentinel, so the walk from the bucket's first entry
      // ends by itself; it runs on the 32-bit high words (32-bit LDS reads and compares) and only a tie there looks at all 64 bits.
      // One induction variable (the byte offset into qhi), made opaque behind the loop: left to itself the optimiser carries four
      // (index, index - 1 and two addresses), each with its own add and copy per step of a loop every lane of the wave waits for
      uint32_t off = (uint32_t)bkt[b] * 4u;
      const uint32_t hh = (uint32_t)(h >> 32);
      const unsigned char* qhiB = (const unsigned char*)qhi;
      uint32_t v = *(const uint32_t*)(qhiB + off);
      while (v < hh) { off += 4u; v = *(const uint32_t*)(qhiB + off); }
      asm volatile("" : "+v"(off));
      int lo = (int)(off >> 2);
      if (v == hh) { while (q[lo] < h) lo++; }
      return (uint32_t)(lo + 1) | (q[lo] == h ? (1u << EF<JB>::MATCH_BIT) : 0u) | ((uint32_t)((int)qs[lo] + 1) << EF<JB>::VOTE_SHIFT);   // query strand + 1
    };
    // the slide ends with the last insert at or before rangeEnd (evictions behind it are never reached, :1340); nextW: the wpos of the
    // record after it in the same contig, else its own (:1387-1390)
    int lastRel = -1, nextW = 0;                                   // lastRel: index of that insert relative to e0
    {
      const uint64_t m = mm_ballot((tailKey & 1u) != 0);
      if (m) { const int l = 63 - (int)__builtin_clzll(m); lastRel = nAll - 64 + l; nextW = (int)((uint32_t)__shfl((int)tailKey, l) >> 1); }
    }
    for (int hiEnd = nAll - 64; hiEnd > nPre && lastRel < 0; hiEnd -= 64) {               // (rare: 64 evictions at the end of a slice)
      const int i = hiEnd - 64 + lane;
      const uint32_t k2 = (i >= nPre && i < hiEnd) ? evKey[in.e0 + i] : 0u;
      const uint64_t m = mm_ballot((k2 & 1u) != 0);
      if (m) { const int l = 63 - (int)__builtin_clzll(m); lastRel = hiEnd - 64 + l; nextW = (int)((uint32_t)__shfl((int)k2, l) >> 1); }
    }
    if (lastRel >= 0) {                                            // everything of the slice behind lastRel is an eviction: go on behind the slice
      const uint64_t m0 = mm_ballot((behindKey & 1u) != 0);
      if (m0) nextW = (int)((uint32_t)__shfl((int)behindKey, (int)__builtin_ctzll(m0)) >> 1);
      else for (int64_t e = in.e0 + nAll + 64; e < ce; e += 64) {
        const uint32_t k2 = e + lane < ce ? evKey[e + lane] : 0u;
        const uint64_t m = mm_ballot((k2 & 1u) != 0);
        if (m) { nextW = (int)((uint32_t)__shfl((int)k2, (int)__builtin_ctzll(m)) >> 1); break; }
      }
    }
    int outN = 0;                                                  // entries written so far (wave-uniform)
    uint32_t preFlags = nOpen + nPre >= preLimit ? 2u : 0u;                    // a cell counts in 12 bits, and no cell can receive more than the pre-load holds: beyond that the exact kernel (32-bit counts) takes the candidate
    int posAcc = cand.rangeStartPos;                               // running position of the delta code (wave-uniform)
    bool tooWide = false;
    const int nEv = lastRel + 1;                                   // events [0, nEv) of the slice are streamed
    const int nCh = nChPre + ((nEv > nPre ? nEv - nPre : 0) + 63) / 64;
    asm volatile("" : "+v"(nKey), "+v"(nAux), "+v"(nHash));        // chunk 0 has arrived (no wait is then needed at the loop's head, where it would also cover the previous chunk's writes)
    for (int ch = 0; ch < nCh; ch++) {
      const uint32_t key = nKey, aux = nAux; const uint64_t h = nHash;
      if (ch + 1 < nCh) loadChunk(ch + 1, nKey, nAux, nHash);       // in flight while this chunk is located
      const bool isIns = (key & 1u) != 0;
      const int pos = (int)(key >> 1);
      const bool slide = ch >= nChPre;                               // wave-uniform
      bool keep, evalIns = false;
      if (!slide) keep = isIns && (int)(aux & 0x7fffffffu) > cand.rangeStartPos && pos >= in.target;   // still open at rangeStart (:1323-1338)
      else {
        const bool live = nPre + (ch - nChPre) * 64 + lane < nEv;
        keep = live && (isIns || h <= qmax);                         // an eviction outside the sketch's range changes nothing
        evalIns = live && isIns;
      }
      uint32_t op = 0;
      if (keep) {
        op = locate(h);
        // vote of a matching insert = query strand x reference strand: a REV record (aux bit 31) mirrors the field around 1
        if (isIns && (aux >> 31)) op = (op & ~(3u << EF<JB>::VOTE_SHIFT)) | ((2u - ((op >> EF<JB>::VOTE_SHIFT) & 3u)) << EF<JB>::VOTE_SHIFT);
        op |= 1u << (!slide ? E_PRE_BIT : (isIns ? E_INS_BIT : E_DEL_BIT));
      }
      if (!slide) {
        // the pre-load (records still open at rangeStart) goes into the cell state, not into the stream: a reference-only hash raises the
        // count of the cell it falls before, a matching one activates its cell with its vote (LDS atomics: the lanes of a chunk hit
        // arbitrary cells; two cells share a dword)
        asm volatile("" : "+v"(nKey), "+v"(nAux), "+v"(nHash));      // (the next chunk's reads are waited for here)
        const uint32_t j = op & EF<JB>::JMASK;
        if (keep && j) {
          const uint32_t sh = (j & 1u) * 16u;
          uint32_t* w = ic + (j >> 1);
          if (op & (1u << EF<JB>::MATCH_BIT)) {
            const uint32_t old = atomicOr(w, 0x1000u << sh);
            if ((old >> sh) & 0x1000u) preFlags |= 1u;               // open twice: only the exact kernel can follow that
            else atomicXor(w, ((1u ^ ((op >> EF<JB>::VOTE_SHIFT) & 3u)) << 13) << sh);   // vote + 1: from 1 (vote 0) to this record's
          } else atomicAdd(w, 1u << sh);                             // (nothing comes back; nPre bounds the count, see preFlags)
        }
        continue;
      }
      const uint64_t mKeep = mm_ballot(keep);
      // delta against the previous evaluated insert (lower lanes of this chunk, else the carry)
      const uint64_t mIns = mm_ballot(evalIns);
      int prevPos = posAcc;
      {
        const uint64_t below = mIns & ((1ull << lane) - 1ull);
        const int src = below ? 63 - (int)__builtin_clzll(below) : lane;
        const int p2 = __shfl(pos, src);
        if (below) prevPos = p2;
      }
      const int delta = evalIns ? pos - prevPos : 0;
      const bool needSkip = evalIns && delta > (int)EF<JB>::MAXDELTA;
      const uint64_t mSkip = mm_ballot(needSkip);                                  // slots before this lane: one per kept event, one more per skip
      const bool bigSkip = mm_ballot(needSkip && (uint32_t)(delta - (int)EF<JB>::MAXDELTA) > E_SKIP_MAX) != 0;
      asm volatile("" : "+v"(nKey), "+v"(nAux), "+v"(nHash));
      if (bigSkip) {
        // a gap that one skip entry (27 bits) cannot carry -- two reference minmers of one candidate more than 2^27 bases apart: the gap
        // is spread over as many skip entries as it takes (the reservation holds one per 2^DELTA_BITS bases of the range).  Round 3
        // failed the batch here.
        const uint32_t extra = needSkip ? (uint32_t)(delta - (int)EF<JB>::MAXDELTA) : 0u;
        const int nSkip = (int)((extra + E_SKIP_MAX - 1u) / E_SKIP_MAX);
        const int mineW = keep ? 1 + nSkip : 0;
        int at = outN + mm_wave_excl_scan(mineW);
        if (keep && at + mineW <= cap) {
          uint32_t left = extra;
          for (int i2 = 0; i2 < nSkip; i2++) { const uint32_t part = left > E_SKIP_MAX ? E_SKIP_MAX : left; out[at++] = (1u << E_SKIP_BIT) | part; left -= part; }
          out[at] = op | ((needSkip ? EF<JB>::MAXDELTA : (uint32_t)delta) << EF<JB>::DELTA_SHIFT);
        } else if (keep) tooWide = true;                                          // cannot happen: see the reservation in k_l2_extents
        outN += mm_wave_sum(mineW);
      } else {
        const int mine = keep ? (needSkip ? 2 : 1) : 0;
        const int at = outN + (int)mm_popc_below(mKeep) + (int)mm_popc_below(mSkip);
        if (keep && at + mine <= cap) {
          if (needSkip) {
            const uint32_t extra = (uint32_t)(delta - (int)EF<JB>::MAXDELTA);
            out[at] = (1u << E_SKIP_BIT) | (extra & E_SKIP_MAX);
            out[at + 1] = op | (EF<JB>::MAXDELTA << EF<JB>::DELTA_SHIFT);
          } else out[at] = op | ((uint32_t)delta << EF<JB>::DELTA_SHIFT);
        }
        outN += __popcll(mKeep) + __popcll(mSkip);
      }
      if (mIns) posAcc = __shfl(pos, 63 - (int)__builtin_clzll(mIns));
    }
    if (lane == 0) {                                               // end marker: carries the wpos behind the last insert
      int at = outN;
      uint32_t delta = lastRel >= 0 ? (uint32_t)(nextW - posAcc) : 0u;
      while (delta > EF<JB>::MAXDELTA) {                           // that record may be far away: as many skips as it takes
        const uint32_t extra = delta - EF<JB>::MAXDELTA > E_SKIP_MAX ? E_SKIP_MAX : delta - EF<JB>::MAXDELTA;
        if (at < cap) out[at] = (1u << E_SKIP_BIT) | extra;
        at++; delta -= extra;
      }
      if (at < cap) out[at] = (1u << E_END_BIT) | (delta << EF<JB>::DELTA_SHIFT);
      else tooWide = true;                                         // cannot happen: the reservation covers every event + skips
    }
    if (mm_ballot(tooWide) && lane == 0) atomicOr(&counters[6], 4ull);
    // ---- the state after the pre-load, in closed form (see L2Init): pivot = number of cells p with count(1) + .. + count(p) <= S (the sums
    // grow with p), pivRank = that sum at the pivot, shared / votes over the active cells up to it
    __threadfence_block();
    {
      int carry = 0, pivot = 0, pivRank = 0, shared = 0, votes = 0;
      for (int p0 = 1; p0 <= S; p0 += 64) {
        const int p = p0 + lane;
        const uint32_t w = p <= S ? (ic[p >> 1] >> ((p & 1) * 16)) & 0xFFFFu : 0u;
        const int cnt = (int)(w & 0xFFFu);
        const int incl = carry + mm_wave_excl_scan(cnt) + cnt;
        const bool ok = p <= S && incl <= S;
        const uint64_t m = mm_ballot(ok);
        if (m) { pivot += (int)__popcll(m); pivRank = __shfl(incl, 63 - (int)__builtin_clzll(m)); }
        const bool act = ok && (w & 0x1000u);
        shared += (int)__popcll(mm_ballot(act));
        votes += mm_wave_sum(act ? (int)((w >> 13) & 3u) - 1 : 0);
        carry = __shfl(incl, 63);
      }
      const uint64_t fl = mm_ballot(preFlags != 0);
      uint32_t flags = 0;
      if (fl) flags = (mm_ballot((preFlags & 1u) != 0) ? 1u : 0u) | (mm_ballot((preFlags & 2u) != 0) ? 2u : 0u);
      // (16-bit stores: writing the row as the dwords it occupies in LDS made the kernel 10 % slower -- profiles/NOTES.md, r13s)
      uint16_t* row = initCells + (size_t)(c - cBase) * (size_t)initStride;
      for (int p = lane; p < S + 2; p += 64) row[p] = (uint16_t)((ic[p >> 1] >> ((p & 1) * 16)) & 0xFFFFu);
      if (lane == 0) initState[c - cBase] = L2Init{pivot | (int32_t)(flags << L2INIT_FLAG_SHIFT), pivRank, shared, votes};
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bit B of x as a lane mask (all ones / zero): one v_bfe_i32, opaque to the optimiser (see k_l2_sweep)
template <int B> __device__ __forceinline__ int mm_bit_mask(uint32_t x) { int m; asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(x), "n"(B)); return m; }

// k_l2_sweep<WIDE>: one lane per candidate.  Per-lane SlideMapper state in LDS, one cell per query position p:
//   num_before_inc (CB bits) | active (1 bit) | strand_vote + 1 (2 bits)
// The LDS that 64 such states need is what limits the waves per CU, and the kernel is latency bound, so the first pass uses
// 8-bit cells (CB = 5: 18 waves/CU at s = 130).  num_before_inc counts the open reference-only hashes between two neighbouring
// query hashes -- about one on average; a candidate where one exceeds 31 is queued and redone with 16-bit cells (CB = 12).
// Cell p of lane l sits at p * LPW + l (16-bit cells of a full wave: half l >> 5 of dword p * 32 + (l & 31)), so that a cell's LDS
// address is one shift-add of its position, the pivot's right neighbour is the same address with an immediate offset, and no address
// arithmetic is left in the per-entry instruction budget -- the kernel is bound by VALU issue, not by LDS: four lanes share a bank of
// the 8-bit layout, which costs the DS pipe a second pass now and then and the SIMDs nothing (round 4; until then every lane owned a
// bank, byte p & 3 of dword (p >> 2) * 64 + l, at five more VALU instructions per cell access).  One cell of padding behind position S
// lets the neighbour be read without clamping (its value is masked when the pivot is at S).
// ---------------------------------------------------------------------------------------------
// LPW lanes of a wave carry a candidate (64; fewer for sketches whose 64 states would not fit a CU's LDS: 32, 16 or 8 -- the other
// lanes leave at once); JB: width of the stream entries' sketch-position field.
template <bool WIDE, int JB, int LPW>
__global__ void __launch_bounds__(64)
k_l2_sweep(int cBase, int nCand, int64_t opsBase, const int32_t* __restrict__ candList, int segLength, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats,
           const int64_t* __restrict__ opOff, const int32_t* __restrict__ opCnt, const uint32_t* __restrict__ ops,
           const int64_t* __restrict__ l1Off, L2Tmp* __restrict__ tmp, int locap, mm_l2_locus* __restrict__ l2, unsigned long long l2Cap,
           int32_t* __restrict__ wideList, int32_t* __restrict__ exactList, int64_t* __restrict__ l2First, int32_t* __restrict__ l2Num,
           unsigned long long* __restrict__ counters /* [0] candidates queued for the exact pass, [4] l2 cursor, [5] overflow, [6] flags, [7] queued for the wide pass */,
           const unsigned long long* __restrict__ nDev /* non-null: the number of candidates (minus cBase) or of list entries lives there */, int listCap,
           const uint16_t* __restrict__ initCells, const L2Init* __restrict__ initState, int initStride, int initBase /* rows of the chunk: candidate - initBase */) {
  typedef typename std::conditional<WIDE, uint16_t, uint8_t>::type CellT;
  constexpr int CB = WIDE ? 12 : 5;
  constexpr uint32_t CMASK = (1u << CB) - 1u;
#define CELL_CNT(x) ((int)((x) & CMASK))
#define CELL_ACT(x) ((int)(((x) >> CB) & 1u))
#define CELL_VOTE(x) ((int)(((x) >> (CB + 1)) & 3u) - 1)
  extern __shared__ __attribute__((aligned(16))) unsigned char cellRaw[];
  CellT* cell = (CellT*)cellRaw;
  const int lane = threadIdx.x;
  const int li = blockIdx.x * LPW + lane;
  if (nDev) {
    const long long nd = (long long)*nDev - (listCap ? 0 : cBase);
    if (listCap && nd > listCap && li == 0) atomicOr(&counters[6], 16ull);   // more listed candidates than this launch covers: the pass is redone with the host's sizing
    nCand = (int)(nd < (listCap ? listCap : nCand) ? nd : (listCap ? listCap : nCand));
  }
  const bool live = lane < LPW && li < nCand;          // (the other lanes of the wave only help with the initial state)
  int cIdx = 0, S = 0, f = 0;
  mm_l1_candidate cand{0, 0, 0, 0, 0};
  if (live) {
    cIdx = candList ? candList[li] : cBase + li;       // candidates [cBase, cBase + nCand), or the listed ones (absolute indices); ops holds the streams from opsBase on
    cand = l1[cIdx]; f = cand.frag; S = stats[f].sketchSize;
  }
  auto lbaseOf = [](int l) -> int { return WIDE && LPW == 64 ? (l & 31) * 2 + (l >> 5) : l; };
  const int lbase = lbaseOf(lane);
#define CELL(p) cell[(p) * LPW + lbase]
  // The state after the pre-load comes ready-made from k_l2_locate (L2Init): every candidate's row of s + 2 16-bit cells is read by the whole
  // wave, 64 consecutive cells at a time, into that candidate's column of the LDS block; a count the 8-bit cells cannot hold flags the
  // candidate for the 16-bit re-run, as an insert into a full cell would have
  int cntOverflow = 0, doubleOpen = 0;
  {
    uint64_t rows = __ballot(live), over = 0;
    while (rows) {
      const int r = (int)__builtin_ctzll(rows); rows &= rows - 1ull;
      const int cR = __shfl(cIdx, r), sR = __shfl(S, r);
      const uint16_t* row = initCells + (size_t)(cR - initBase) * (size_t)initStride;
      const int lb = lbaseOf(r);
      bool tooBig = false;
      for (int p = (int)threadIdx.x; p < sR + 2; p += 64) {
        const uint32_t w = row[p];
        if (WIDE) cell[p * LPW + lb] = (CellT)w;
        else {
          const uint32_t cnt = w & 0xFFFu;
          tooBig |= cnt > CMASK;
          cell[p * LPW + lb] = (CellT)((cnt & CMASK) | (((w >> 12) & 1u) << CB) | (((w >> 13) & 3u) << (CB + 1)));
        }
      }
      if (__ballot(tooBig)) over |= 1ull << r;
    }
    __threadfence_block();
    if (!live) return;
    if ((over >> lane) & 1ull) cntOverflow = -1;
  }
  const L2Init ist = initState[cIdx - initBase];
  const int iflags = (int)((uint32_t)ist.pivot >> L2INIT_FLAG_SHIFT);
  if (iflags & 1) doubleOpen = -1;                     // a query hash open twice in the pre-load: k_l2_sweep_exact
  if (iflags & 2) cntOverflow = -1;                    // a pre-load too long for 12-bit counts: likewise (through the 16-bit re-run's own overflow)
  const uint4* src = (const uint4*)(ops + (opOff[cIdx] - opsBase));
  const int nSteps = opCnt[cIdx] / E_STEP;             // 16 entries = 4 x 16 bytes per step
  int posAcc = cand.rangeStartPos;                     // running position of the delta code
  int pivot = ist.pivot & ((1 << L2INIT_FLAG_SHIFT) - 1), pivRank = ist.pivRank, shared = ist.shared, votes = ist.votes;

  // Lane masks.  The 64 lanes of a wave are at different candidates and take different cases at every entry, so the cases are
  // not branches (measured 5x slower) but all-ones / zero integers combined with and/or/add: a kind bit of the entry becomes
  // a mask with one v_bfe_i32, a comparison of two small non-negative numbers with a subtract and an arithmetic shift, and
  // "x += c ? a : 0" is "x += a & m".  These are 32-bit-encoded VALU instructions, which issue about twice as fast on this
  // part as the v_cmp + v_cndmask pairs (64-bit encodings) the bool form compiles to.  The mask
  // producers are inline asm so that the optimiser does not turn "mask & value" back into compare + select.
#define BITM(x, b) mm_bit_mask<b>((uint32_t)(x))
  auto neg = [](int x) -> int { int m; asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(x)); return m; };   // x < 0 ? ~0 : 0
  auto sel = [](int m, int a, int b2) -> int { return (a & m) | (b2 & ~m); };                               // v_bfi_b32

  // SlideMapper::insert_minmer / delete_minmer (slidingMap.hpp:125-211): the four cases (insert or delete x hash matches a
  // query hash or not) as masks IM, IN, DM, DN.  The cell of the hash, of the pivot and of its right neighbour are read up
  // front, independent of the case.  insM: the entry inserts (insert or pre-load), delM: it evicts.
  auto apply = [&](uint32_t e, int insM, int delM) {
    const int j = (int)(e & EF<JB>::JMASK) & (insM | delM);   // 0: no hash / hash beyond the query sketch -> no effect (cell 0 is a dummy)
    const int valid = neg(-j) & ~cntOverflow;          // after a counter overflow the lane only idles to the end
    const int mt = BITM(e, EF<JB>::MATCH_BIT);
    const int ltS = neg(pivot - S);                    // pivot + 1 <= S
    const int pn = pivot + 1;                          // at pivot == S: the padding cell, masked by ltS below
    const uint32_t cw = CELL(j), pw = CELL(pivot), nw = CELL(pn);
    const int vi = valid & insM, vd = valid & delM;
    const int IM = vi & mt, IN = vi & ~mt, DM = vd & mt, DN = vd & ~mt;
    const int vf = (int)((e >> EF<JB>::VOTE_SHIFT) & 3u);   // vote + 1; a query hash normally has one open reference window at a time (windowLen == 0):
    doubleOpen |= IM & BITM(cw, CB);                   // the 2-bit vote relies on it; a candidate that violates it is handed to k_l2_sweep_exact
    cntOverflow |= IN & neg((int)(CMASK - 1u) - (int)(cw & CMASK));   // the counter of this cell is full: redo the candidate with wide cells
    const int repl = (int)(cw & CMASK) | sel(IM, (int)((1u << CB) | ((uint32_t)vf << (CB + 1))), (int)(1u << (CB + 1)));
    const int ncw = sel(IM | DM, repl, (int)cw - IN + DN);            // IN: count + 1, DN: count - 1 (the masks are -1)
    CELL(j) = (CellT)ncw;
    const int ip = ~neg(pivot - j);                    // j <= pivot
    const int IMp = IM & ip, DMp = DM & ip;
    shared += DMp - IMp;
    votes += ((vf - 1) & IMp) - (CELL_VOTE(cw) & DMp);
    pivRank += (DN & ip) - (IN & ip);
    // insert of a non-shared hash can push the pivot one cell left (:155-160)
    const int pwp = sel(neg((pivot ^ j) - 1), ncw, (int)pw);          // pivot == j: the cell just written
    const int left = IN & neg(S - pivRank);                           // pivRank > S
    // delete of a non-shared hash can let it move one cell right (:201-207)
    const int nwp = sel(neg((pn ^ j) - 1), ncw, (int)nw);
    const int right = DN & ltS & ~neg(S - pivRank - CELL_CNT(nwp));   // pivRank + count(next) <= S
    shared += (CELL_ACT(nwp) & right) - (CELL_ACT(pwp) & left);
    votes += (CELL_VOTE(nwp) & right) - (CELL_VOTE(pwp) & left);
    pivRank += (CELL_CNT(nwp) & right) - (CELL_CNT(pwp) & left);
    pivot += left - right;
  };

  // best-position bookkeeping (:1376-1449)
  int bestShared = 1, inRun = 0;
  int curStart = 0, curEnd = 0, curShared = 0;
  int nFlushed = 0, havePend = 0; L2Tmp pend{0, 0, 0, 0};
  L2Tmp* mySlots = tmp + (size_t)cIdx * locap;
  bool slotOverflow = false;
  auto close_run = [&](int strand) {                   // :1417-1426 / :1440-1449
    if (!havePend || pend.end + segLength < curStart) {
      if (havePend) { if (nFlushed < locap) mySlots[nFlushed] = pend; else slotOverflow = true; nFlushed++; }
      pend.start = curStart; pend.end = curEnd; pend.shared = curShared; pend.strand = strand; havePend = -1;
    } else {
      pend.end = curEnd;
    }
  };
  // the evaluation of insert i needs the wpos of record i+1, so it is carried until the next insert / end entry arrives
  // lastVotes = strand_votes right after the most recent insert (pre-load included): the reference samples it before the
  // evictions that precede the next insert (:1342)
  int evalPending = 0, evW = 0, evShared = 0, evPrevVotes = 0, lastVotes = votes;      // (votes right after the last pre-load insert)
  // the three outcomes of an evaluation (:1376-1430) as masks; only the closing of a run -- rare -- is a branch
  auto evaluate = [&](int on, int nextW) {
    const int gt = on & neg(bestShared - evShared), lt = on & neg(evShared - bestShared), ge = on & ~lt;
    if (lt & inRun) { curEnd = nextW; close_run(evPrevVotes >= 0 ? 1 : -1); curStart = 0; curEnd = 0; curShared = 0; }
    const int startNew = gt | (ge & ~inRun);
    nFlushed &= ~gt; havePend &= ~gt;                  // l2_vec_out.clear()
    bestShared = sel(gt, evShared, bestShared);
    curShared = sel(startNew, evShared, curShared);
    curStart = sel(startNew, evW, curStart);
    curEnd = sel(ge, nextW, curEnd);
    inRun = sel(on, ge, inRun);
  };

  int done = 0;
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
    for (int k = 0; k < E_STEP; k++) {
      const uint4 v = cur[k >> 2];
      const uint32_t eRaw = (k & 3) == 0 ? v.x : (k & 3) == 1 ? v.y : (k & 3) == 2 ? v.z : v.w;
      const uint32_t e = eRaw & (uint32_t)~done;         // a finished lane reads on behind its end marker: no kind bit, no effect
      // straight-line per entry: the entry kinds are masks, not branches
      const int mIns = BITM(e, 27), mDel = BITM(e, 28), mEnd = BITM(e, 30);              // (no pre-load entries: the state starts behind them)
      const int mSkip = neg((int)e);
      const int ie = mIns | mEnd;
      posAcc += ((int)((e >> EF<JB>::DELTA_SHIFT) & EF<JB>::MAXDELTA) & ie) | ((int)(e & E_SKIP_MAX) & mSkip);
      const int wpos = posAcc;
      evaluate(ie & evalPending, wpos);
      evalPending = (evalPending & ~ie) | mIns;
      done |= mEnd;
      evPrevVotes = sel(mIns, lastVotes, evPrevVotes);
      apply(e, mIns, mDel);
      lastVotes = sel(mIns, votes, lastVotes);
      evW = sel(mIns, wpos, evW); evShared = sel(mIns, shared, evShared);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nxt[k];
  }
  if (inRun) close_run(votes >= 0 ? 1 : -1);
#undef CELL
#undef CELL_CNT
#undef CELL_ACT
#undef CELL_VOTE
  int total = nFlushed + (havePend ? 1 : 0);
  if (cntOverflow) {
    total = 0;
    if (WIDE) exactList[atomicAdd(&counters[0], 1ull)] = cIdx;   // > 4094 open reference-only hashes between two query hashes: the literal kernel counts in 32 bits
    else wideList[atomicAdd(&counters[7], 1ull)] = cIdx;
  }
  if (doubleOpen && !cntOverflow) {                    // a query hash with two reference windows open at once: the 2-bit vote cell cannot hold it;
    total = 0; slotOverflow = false;                   // the candidate is redone by k_l2_sweep_exact
    exactList[atomicAdd(&counters[0], 1ull)] = cIdx;
  }
  if (slotOverflow && !cntOverflow) { atomicOr(&counters[6], 1ull); total = 0; }
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
  l2First[cIdx] = (int64_t)(wbase + (unsigned long long)(incl - total)); l2Num[cIdx] = total;   // where k_l2_select finds this candidate's loci
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
// k_l2_sweep_exact: the SlideMapper sweep of one candidate per thread, literally (slidingMap.hpp:125-211), for the candidates the fast
// kernels hand over: an index in which two windows of one hash overlap (addMinmers only removes ADJACENT duplicates,
// commonFunc.hpp:560; an index written by another program may hold anything) makes a query hash active twice.  The reference's state
// then keeps `active` a boolean, accumulates strand_vote, and counts sharedSketchElements once per insert and once per delete -- which
// is what this kernel does, cell by cell, with the state in global memory (count, active, accumulated vote per query position).
// Rare by construction, so nothing here is tuned.
// ---------------------------------------------------------------------------------------------
struct ExactCell { int32_t cnt; int16_t vote; int16_t active; };
__global__ void __launch_bounds__(64)
k_l2_sweep_exact(int jb, int nList, int64_t opsBase, const int32_t* __restrict__ list, int segLength, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats,
                 const int64_t* __restrict__ opOff, const int32_t* __restrict__ opCnt, const uint32_t* __restrict__ ops,
                 const int64_t* __restrict__ l1Off, ExactCell* __restrict__ cells, int cellStride, L2Tmp* __restrict__ tmp, int locap,
                 mm_l2_locus* __restrict__ l2, unsigned long long l2Cap, int64_t* __restrict__ l2First, int32_t* __restrict__ l2Num,
                 unsigned long long* __restrict__ counters, const unsigned long long* __restrict__ nDev,
                 const L2Info* __restrict__ info, int s, const uint64_t* __restrict__ qHash, const int8_t* __restrict__ qStrand,
                 const uint64_t* __restrict__ skHash, const int8_t* __restrict__ skStrand,
                 const uint32_t* __restrict__ evKey, const uint32_t* __restrict__ evAux, const uint64_t* __restrict__ evHash,
                 const uint32_t* __restrict__ opKey, const uint32_t* __restrict__ opAux, const uint64_t* __restrict__ opHash) {
  const int li = blockIdx.x * 64 + threadIdx.x;
  if (nDev) {                                          // steady state: the launch covers nList entries, the list's length is on the device
    const long long nd = (long long)*nDev;
    if (nd > nList && li == 0) atomicOr(&counters[6], 16ull);
    if (nd < nList) nList = (int)nd;
  }
  if (li >= nList) return;
  const int cIdx = list[li];
  const mm_l1_candidate cand = l1[cIdx];
  const int f = cand.frag;
  const int S = stats[f].sketchSize;
  const uint32_t jmask = (1u << jb) - 1u; const int deltaShift = jb + 3; const uint32_t maxDelta = (1u << (27 - deltaShift)) - 1u;
  ExactCell* cell = cells + (size_t)li * cellStride;
  cell[0] = ExactCell{0, 0, 0};
  for (int p = 1; p <= S; p++) cell[p] = ExactCell{1, 0, 0};                   // SlideMapper::init (:103-121)
  int pivot = S, pivRank = S, shared = 0, votes = 0;
  const uint32_t* src = ops + (opOff[cIdx] - opsBase);
  const int nEnt = opCnt[cIdx];
  int posAcc = cand.rangeStartPos;
  int bestShared = 1; bool inRun = false;
  int curStart = 0, curEnd = 0, curShared = 0;
  int nFlushed = 0; bool havePend = false; L2Tmp pend{0, 0, 0, 0};
  L2Tmp* mySlots = tmp + (size_t)cIdx * locap;
  bool slotOverflow = false;
  auto close_run = [&](int strand) {                                             // computeMap.hpp:1417-1426 / :1440-1449
    if (!havePend || pend.end + segLength < curStart) {
      if (havePend) { if (nFlushed < locap) mySlots[nFlushed] = pend; else slotOverflow = true; nFlushed++; }
      pend.start = curStart; pend.end = curEnd; pend.shared = curShared; pend.strand = strand; havePend = true;
    } else pend.end = curEnd;
  };
  bool evalPending = false; int evW = 0, evShared = 0, evPrevVotes = 0, lastVotes = 0;
  // insert_minmer (slidingMap.hpp:125-165) for the hash at 1-based sketch position j (0: beyond the sketch, no effect)
  auto insert = [&](int j, bool match, int vote) {
    if (j <= 0) return;
    ExactCell c = cell[j];
    if (match) {
      c.active = 1; c.vote = (int16_t)(c.vote + vote);
      if (j <= pivot) { shared++; votes += c.vote; }
      cell[j] = c;
    } else {
      c.cnt++; cell[j] = c;
      if (j <= pivot) pivRank++;
      if (pivRank > S) { const ExactCell pc = cell[pivot]; shared -= pc.active; votes -= pc.vote; pivRank -= pc.cnt; pivot--; }
    }
  };
  {
    // The pre-load (computeMap.hpp:1323-1338) is not in the stream (the fast kernels get its result in closed form, L2Init): it is replayed
    // here from the index, record by record in index order -- the block's list of open records, then the events before rangeStart --, with
    // k_l2_locate's rule for what is still open at rangeStart and a binary search of the query sketch in place of its LDS table
    const L2Info in = info[cIdx];
    const bool raw = in.sketch < 0;
    const uint64_t* qh = (raw ? skHash : qHash) + (size_t)f * s;
    const int8_t* qv = (raw ? skStrand : qStrand) + (size_t)f * s;
    const uint64_t qmax = S > 0 ? qh[S - 1] : 0ull;
    auto one = [&](uint32_t key, uint32_t aux, uint64_t h) {
      const bool isInsert = (key & 1u) != 0; const int pos = (int)(key >> 1);
      if (!(isInsert && (int)(aux & 0x7fffffffu) > cand.rangeStartPos && pos >= in.target)) return;
      if (S > 0 && h <= qmax) {
        int lo = 0, hi = S;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (qh[mid] < h) lo = mid + 1; else hi = mid; }
        const bool match = qh[lo] == h;
        insert(lo + 1, match, (aux >> 31) ? -(int)qv[lo] : (int)qv[lo]);        // vote = query strand x reference strand
      }
      lastVotes = votes;
    };
    for (int i = 0; i < in.nOpen; i++) one(opKey[in.open0 + i], opAux[in.open0 + i], opHash[in.open0 + i]);
    for (int i = 0; i < in.nPre; i++) one(evKey[in.e0 + i], evAux[in.e0 + i], evHash[in.e0 + i]);
  }
  for (int i = 0; i < nEnt; i++) {
    const uint32_t e = src[i];
    const bool isIns = (e >> E_INS_BIT) & 1u, isDel = (e >> E_DEL_BIT) & 1u, isPre = (e >> E_PRE_BIT) & 1u, isEnd = (e >> E_END_BIT) & 1u, isSkip = (e >> E_SKIP_BIT) & 1u;
    if (isSkip) { posAcc += (int)(e & E_SKIP_MAX); continue; }
    if (isIns || isEnd) posAcc += (int)((e >> deltaShift) & maxDelta);
    const int wpos = posAcc;
    if ((isIns || isEnd) && evalPending) {                                       // the evaluation behind the previous insert (:1376-1430): it needed this wpos
      if (evShared > bestShared) {
        nFlushed = 0; havePend = false; bestShared = evShared;
        curShared = evShared; curStart = evW; curEnd = wpos; inRun = true;
      } else if (evShared == bestShared) {
        if (!inRun) { curShared = evShared; curStart = evW; }
        curEnd = wpos; inRun = true;
      } else {
        if (inRun) { curEnd = wpos; close_run(evPrevVotes >= 0 ? 1 : -1); curStart = curEnd = curShared = 0; }
        inRun = false;
      }
      evalPending = false;
    }
    if (isEnd) break;
    if (isIns) { evalPending = true; evPrevVotes = lastVotes; }
    const int j = (int)(e & jmask);
    if (j > 0 && (isIns || isPre || isDel)) {
      const bool match = (e >> jb) & 1u;
      ExactCell c = cell[j];
      if (isIns || isPre) {                                                      // insert_minmer (:125-165)
        if (match) {
          c.active = 1; c.vote = (int16_t)(c.vote + ((int)((e >> (jb + 1)) & 3u) - 1));
          if (j <= pivot) { shared++; votes += c.vote; }
          cell[j] = c;
        } else {
          c.cnt++; cell[j] = c;
          if (j <= pivot) pivRank++;
          if (pivRank > S) { const ExactCell pc = cell[pivot]; shared -= pc.active; votes -= pc.vote; pivRank -= pc.cnt; pivot--; }
        }
      } else {                                                                   // delete_minmer (:171-211)
        if (match) {
          if (j <= pivot) { shared--; votes -= c.vote; }
          c.active = 0; c.vote = 0; cell[j] = c;
        } else {
          c.cnt--; cell[j] = c;
          if (j <= pivot) pivRank--;
          if (pivot + 1 <= S && pivRank + cell[pivot + 1].cnt <= S) { pivot++; const ExactCell pc = cell[pivot]; shared += pc.active; votes += pc.vote; pivRank += pc.cnt; }
        }
      }
    }
    if (isIns || isPre) lastVotes = votes;
    if (isIns) { evW = wpos; evShared = shared; }
  }
  if (inRun) close_run(votes >= 0 ? 1 : -1);
  int total = nFlushed + (havePend ? 1 : 0);
  if (slotOverflow) { atomicOr(&counters[6], 1ull); total = 0; }
  unsigned long long base = 0;
  if (total > 0) {
    base = atomicAdd(&counters[4], (unsigned long long)total);
    if (base + (unsigned long long)total > l2Cap) { atomicOr(&counters[5], 1ull); return; }
  }
  l2First[cIdx] = (int64_t)base; l2Num[cIdx] = total;
  const int candLocal = (int)(cIdx - l1Off[f]);
  for (int k = 0; k < total; k++) {
    const L2Tmp t = (k < nFlushed) ? mySlots[k] : pend;
    mm_l2_locus o;
    o.frag = f; o.cand = candLocal; o.seqId = cand.seqId; o.optimalStart = t.start; o.optimalEnd = t.end;
    o.meanOptimalPos = (t.start + t.end) / 2; o.sharedSketchSize = t.shared; o.strand = t.strand;
    l2[base + k] = o;
  }
}


// ---------------------------------------------------------------------------------------------
// windowLen != 0 (--noSplit with a read longer than segLength; computeMap.hpp:1276-1451 with Q.len > segLength): the L2 stage literally, one
// thread per candidate.  minmerIndex is walked as the insert events of the contig's stream (they are its records, in index order); the
// open records sit in a heap ordered by wpos_end with libstdc++'s element movements (std::push_heap / std::pop_heap, mm_heap.h); a
// count of open windows per hash (hash_to_freq, :1310) decides, as in the reference, whether a record enters the SlideMapper and whether
// the position is evaluated at all -- including the reference's way of retiring a record whose hash was counted more than once
// (:1344-1357: the front is decremented until its count reaches zero, then popped).  State per candidate in HBM scratch: SlideMapper
// cells, the heap, an open-addressing table for the counts.  A slow path by design: correctness for the one mode the fast kernels
// cannot express; split mode never comes here.
// ---------------------------------------------------------------------------------------------
struct WinExt { int64_t e0, e1; };                               // events [e0, e1) of the contig: from lower_bound(rangeStart - segLength - 1) to the last wpos <= rangeEnd + windowLen
__global__ void __launch_bounds__(256)
k_l2_window_extents(int nCand, int segLength, const mm_l1_candidate* __restrict__ l1, const DFrag* __restrict__ frags, const uint32_t* __restrict__ evKey,
                    const int64_t* __restrict__ contigOff, WinExt* __restrict__ ext, int32_t* __restrict__ cntH, int32_t* __restrict__ cntT) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nCand) return;
  const mm_l1_candidate cand = l1[c];
  int W = frags[cand.frag].len - segLength; if (W < 0) W = 0;
  const int64_t cb = contigOff[cand.seqId], ce = contigOff[cand.seqId + 1];
  auto lowerIn = [&](int64_t lo, int64_t hi, uint64_t key) { while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((uint64_t)evKey[mid] < key) lo = mid + 1; else hi = mid; } return lo; };
  const long long target = (long long)cand.rangeStartPos - segLength - 1;
  const int64_t e0 = target <= 0 ? cb : lowerIn(cb, ce, (uint64_t)target * 2ull);
  const long long last = (long long)cand.rangeEndPos + W;
  const int64_t e1 = last < 0 ? e0 : lowerIn(e0, ce, (uint64_t)last * 2ull + 2ull);
  ext[c] = WinExt{e0, e1 < e0 ? e0 : e1};
  const int64_t n = (e1 < e0 ? 0 : e1 - e0) + 2;
  int64_t t = 16; while (t < 2 * n) t <<= 1;
  cntH[c] = (int32_t)n; cntT[c] = (int32_t)t;
}

__global__ void __launch_bounds__(64)
k_l2_window(int nCand, int s, int segLength, const mm_l1_candidate* __restrict__ l1, const mm_frag_stats* __restrict__ stats, const DFrag* __restrict__ frags,
            const uint64_t* __restrict__ qHash, const int8_t* __restrict__ qStrand, const uint64_t* __restrict__ skHash, const int8_t* __restrict__ skStrand,
            const uint32_t* __restrict__ evKey, const uint32_t* __restrict__ evAux, const uint64_t* __restrict__ evHash, const int64_t* __restrict__ contigOff,
            const WinExt* __restrict__ ext, const int64_t* __restrict__ offH, const int64_t* __restrict__ offT, const int32_t* __restrict__ cntT,
            int32_t* __restrict__ heapAll, uint64_t* __restrict__ tabKeys, int32_t* __restrict__ tabVals, ExactCell* __restrict__ cells,
            const int64_t* __restrict__ l1Off, L2Tmp* __restrict__ tmp, int locap, mm_l2_locus* __restrict__ l2, unsigned long long l2Cap,
            int64_t* __restrict__ l2First, int32_t* __restrict__ l2Num, unsigned long long* __restrict__ counters) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nCand) return;
  const mm_l1_candidate cand = l1[c];
  const int f = cand.frag;
  const mm_frag_stats fst = stats[f];
  const int S = fst.sketchSize;
  int W = frags[f].len - segLength; if (W < 0) W = 0;
  const bool raw = fst.rawSketchSize == fst.sketchSize;                         // no seed was removed: the sketch is the raw one (k_lookup_l1)
  const uint64_t* q = (raw ? skHash : qHash) + (size_t)f * s;                  // Q.minmerTableQuery, ascending
  const int8_t* qs = (raw ? skStrand : qStrand) + (size_t)f * s;
  ExactCell* cell = cells + (size_t)c * (s + 1);
  cell[0] = ExactCell{0, 0, 0};
  for (int p = 1; p <= S; p++) cell[p] = ExactCell{1, 0, 0};                   // SlideMapper::init (slidingMap.hpp:103-121)
  int pivot = S, pivRank = S, shared = 0, votes = 0;
  int32_t* heap = heapAll + offH[c]; int nHeap = 0;                             // event indices relative to e0
  uint64_t* tk = tabKeys + offT[c]; int32_t* tv = tabVals + offT[c];
  const uint32_t tmask = (uint32_t)cntT[c] - 1u;
  for (uint32_t i = 0; i <= tmask; i++) { tk[i] = ~0ull; tv[i] = 0; }
  const WinExt X = ext[c];
  const int64_t ce = contigOff[cand.seqId + 1];
  auto wposOf = [&](int64_t e) { return (int)(evKey[e] >> 1); };
  auto wendOf = [&](int64_t e) { return (int)(evAux[e] & 0x7fffffffu); };
  auto freqOf = [&](uint64_t h) -> int32_t* {                                    // hash_to_freq[h] (created at 0 on first access, like operator[])
    uint32_t i = (uint32_t)(h * 0x9E3779B97F4A7C15ull >> 40) & tmask;
    while (tk[i] != h) { if (tk[i] == ~0ull) { tk[i] = h; break; } i = (i + 1) & tmask; }
    return &tv[i];
  };
  auto locate = [&](uint64_t h, bool& match) -> int {                            // 1-based lower_bound in the sketch; 0: beyond its last hash
    int lo = 0, hi = S;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (q[mid] < h) lo = mid + 1; else hi = mid; }
    if (lo >= S) { match = false; return 0; }
    match = q[lo] == h; return lo + 1;
  };
  auto insertMinmer = [&](int64_t e) {                                           // slidingMap.hpp:125-165
    bool match; const int j = locate(evHash[e], match);
    if (j == 0) return;
    ExactCell x = cell[j];
    if (match) {
      x.active = 1; x.vote = (int16_t)(x.vote + (int)qs[j - 1] * ((evAux[e] >> 31) ? -1 : 1));
      cell[j] = x;
      if (j <= pivot) { shared++; votes += x.vote; }
    } else {
      x.cnt++; cell[j] = x;
      if (j <= pivot) pivRank++;
      if (pivRank > S) { const ExactCell pc = cell[pivot]; shared -= pc.active; votes -= pc.vote; pivRank -= pc.cnt; pivot--; }
    }
  };
  auto deleteMinmer = [&](int64_t e) {                                           // slidingMap.hpp:171-211
    bool match; const int j = locate(evHash[e], match);
    if (j == 0) return;
    ExactCell x = cell[j];
    if (match) {
      if (j <= pivot) { shared--; votes -= x.vote; }
      x.active = 0; x.vote = 0; cell[j] = x;
    } else {
      x.cnt--; cell[j] = x;
      if (j <= pivot) pivRank--;
      if (pivot + 1 <= S && pivRank + cell[pivot + 1].cnt <= S) { pivot++; const ExactCell pc = cell[pivot]; shared += pc.active; votes += pc.vote; pivRank += pc.cnt; }
    }
  };
  auto later = [&](int32_t a, int32_t b) { return wendOf(X.e0 + a) > wendOf(X.e0 + b); };   // heap_cmp (:1299): min-heap on wpos_end
  auto nextIns = [&](int64_t e) { while (e < ce && !(evKey[e] & 1u)) e++; return e; };       // minmerIndex records = insert events
  // wpos of the record behind `e` in the same contig, else its own (:1387-1390)
  auto nextWpos = [&](int64_t e) { const int64_t n = nextIns(e + 1); return n < ce ? wposOf(n) : wposOf(e); };

  int bestShared = 1; bool inRun = false;
  int curStart = 0, curEnd = 0, curShared = 0;
  int nFlushed = 0; bool havePend = false; L2Tmp pend{0, 0, 0, 0};
  L2Tmp* mySlots = tmp + (size_t)c * locap;
  bool slotOverflow = false;
  auto close_run = [&](int strand) {                                             // :1417-1426 / :1440-1449
    if (!havePend || pend.end + segLength < curStart) {
      if (havePend) { if (nFlushed < locap) mySlots[nFlushed] = pend; else slotOverflow = true; nFlushed++; }
      pend.start = curStart; pend.end = curEnd; pend.shared = curShared; pend.strand = strand; havePend = true;
    } else pend.end = curEnd;
  };

  int64_t it = nextIns(X.e0);
  while (it < ce && wposOf(it) < cand.rangeStartPos) {                           // set up the window (:1323-1338)
    if (wendOf(it) > cand.rangeStartPos) {
      int32_t* fq = freqOf(evHash[it]);
      if (W > 0) (*fq)++;
      if (W == 0 || *fq == 1) {
        heap[nHeap] = (int32_t)(it - X.e0); nHeap++;
        mm_heap_push(heap, nHeap - 1, 0, heap[nHeap - 1], later);
        insertMinmer(it);
      }
    }
    it = nextIns(it + 1);
  }
  while (it < ce && (long long)wposOf(it) <= (long long)cand.rangeEndPos + W) {  // the slide (:1340-1434)
    const int prevVotes = votes;
    while (nHeap > 0 && wendOf(X.e0 + heap[0]) <= wposOf(it) - W) {
      const int64_t fr = X.e0 + heap[0];
      int32_t* fq = freqOf(evHash[fr]);
      if (W > 0) (*fq)--;
      if (W == 0 || *fq == 0) {
        deleteMinmer(fr);
        mm_pop_heap(heap, nHeap, later); nHeap--;
      }
    }
    int32_t* fq = freqOf(evHash[it]);
    if (W > 0) (*fq)++;
    if (W == 0 || *fq == 1) {
      insertMinmer(it);
      heap[nHeap] = (int32_t)(it - X.e0); nHeap++;
      mm_heap_push(heap, nHeap - 1, 0, heap[nHeap - 1], later);
    } else { it = nextIns(it + 1); continue; }
    if (shared > bestShared) {
      nFlushed = 0; havePend = false;
      inRun = true; bestShared = shared; curShared = shared;
      curStart = wposOf(it);                                                     // (no "- windowLen" here in the reference, :1384)
      curEnd = nextWpos(it) - W;
    } else if (shared == bestShared) {
      if (!inRun) { curShared = shared; curStart = wposOf(it) - W; }
      inRun = true;
      curEnd = nextWpos(it) - W;
    } else {
      if (inRun) { curEnd = nextWpos(it) - W; close_run(prevVotes >= 0 ? 1 : -1); curStart = curEnd = curShared = 0; }
      inRun = false;
    }
    it = nextIns(it + 1);
  }
  if (inRun) close_run(votes >= 0 ? 1 : -1);
  int total = nFlushed + (havePend ? 1 : 0);
  if (slotOverflow) { atomicOr(&counters[6], 1ull); total = 0; }
  unsigned long long base = 0;
  if (total > 0) {
    base = atomicAdd(&counters[4], (unsigned long long)total);
    if (base + (unsigned long long)total > l2Cap) { atomicOr(&counters[5], 1ull); return; }
  }
  l2First[c] = (int64_t)base; l2Num[c] = total;
  const int candLocal = (int)(c - l1Off[f]);
  for (int k = 0; k < total; k++) {
    const L2Tmp t = (k < nFlushed) ? mySlots[k] : pend;
    mm_l2_locus o;
    o.frag = f; o.cand = candLocal; o.seqId = cand.seqId; o.optimalStart = t.start; o.optimalEnd = t.end;
    o.meanOptimalPos = (t.start + t.end) / 2; o.sharedSketchSize = t.shared; o.strand = t.strand;
    l2[base + k] = o;
  }
}

static int mm_launch_l2_window(mm_ctx* c, unsigned long long* cnt) {
  const DeviceIndex& I = c->idx;
  const int s = c->P.sketchSize;
  const int nC = (int)c->nL1;
  MM_HIP(c, c->dWinExt.ensure((size_t)nC * sizeof(WinExt) + 64));
  MM_HIP(c, c->dWinCntH.ensure((size_t)nC * 4 + 64)); MM_HIP(c, c->dWinCntT.ensure((size_t)nC * 4 + 64));
  MM_HIP(c, c->dWinOffH.ensure((size_t)nC * 8 + 64)); MM_HIP(c, c->dWinOffT.ensure((size_t)nC * 8 + 64));
  MM_HIP(c, c->dL2First.ensure((size_t)nC * 8 + 64)); MM_HIP(c, c->dL2Num.ensure((size_t)nC * 4 + 64));
  int64_t totH = 0, totT = 0;
  {
    KernelTimer t(c, MM_K_L2_LOCATE);
    hipLaunchKernelGGL(k_l2_window_extents, dim3((nC + 255) / 256), dim3(256), 0, c->stream, nC, c->P.segLength, c->dL1.as<mm_l1_candidate>(), c->dFrags.as<DFrag>(),
                       I.evKey.as<uint32_t>(), I.contigOff.as<int64_t>(), c->dWinExt.as<WinExt>(), c->dWinCntH.as<int32_t>(), c->dWinCntT.as<int32_t>());
    MM_HIP(c, hipGetLastError());
    int rc = mm_scan_i32_to_i64(c, nC, c->dWinCntH.as<int32_t>(), c->dWinOffH.as<int64_t>(), &totH); if (rc != MM_OK) return rc;
    rc = mm_scan_i32_to_i64(c, nC, c->dWinCntT.as<int32_t>(), c->dWinOffT.as<int64_t>(), &totT); if (rc != MM_OK) return rc;
  }
  if ((size_t)totT * 12 + (size_t)totH * 4 > ((size_t)96 << 30)) { c->err = "windowLen != 0: the candidates of this batch need more than 96 GiB of scratch; use smaller batches (MASHMAP_HIP_BATCH_MBP)"; return MM_ERR_CAPACITY; }
  MM_HIP(c, c->dWinHeap.ensure((size_t)totH * 4 + 64)); MM_HIP(c, c->dWinKeys.ensure((size_t)totT * 8 + 64)); MM_HIP(c, c->dWinVals.ensure((size_t)totT * 4 + 64));
  MM_HIP(c, c->dL2Cells.ensure((size_t)nC * (size_t)(s + 1) * sizeof(ExactCell) + 64));
  if (c->l2Cap < c->nL1 * 2 + 1024) c->l2Cap = c->nL1 * 2 + 1024;
  unsigned long long hc[8];
  int locap = MM_LOCAP0;
  for (int attempt = 0; attempt < 24; attempt++) {
    MM_HIP(c, c->dL2.ensure(c->l2Cap * sizeof(mm_l2_locus) + 64));
    MM_HIP(c, c->dL2Tmp.ensure((size_t)nC * locap * sizeof(L2Tmp) + 64));
    MM_HIP(c, hipMemsetAsync(cnt + 4, 0, 24, c->stream));
    {
      KernelTimer t(c, MM_K_L2);
      hipLaunchKernelGGL(k_l2_window, dim3((nC + 63) / 64), dim3(64), 0, c->stream, nC, s, c->P.segLength, c->dL1.as<mm_l1_candidate>(), c->dStats.as<mm_frag_stats>(),
                         c->dFrags.as<DFrag>(), c->dQHash.as<uint64_t>(), c->dQStrand.as<int8_t>(), c->dSkHash.as<uint64_t>(), c->dSkStrand.as<int8_t>(),
                         I.evKey.as<uint32_t>(), I.evAux.as<uint32_t>(), I.evHash.as<uint64_t>(), I.contigOff.as<int64_t>(), c->dWinExt.as<WinExt>(),
                         c->dWinOffH.as<int64_t>(), c->dWinOffT.as<int64_t>(), c->dWinCntT.as<int32_t>(), c->dWinHeap.as<int32_t>(), c->dWinKeys.as<uint64_t>(),
                         c->dWinVals.as<int32_t>(), c->dL2Cells.as<ExactCell>(), c->dL1Off.as<int64_t>(), c->dL2Tmp.as<L2Tmp>(), locap, c->dL2.as<mm_l2_locus>(),
                         (unsigned long long)c->l2Cap, c->dL2First.as<int64_t>(), c->dL2Num.as<int32_t>(), cnt);
      MM_HIP(c, hipGetLastError());
    }
    MM_HIP(c, hipMemcpyAsync(hc, cnt, 64, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
    if (hc[6] & 1ull) { if ((size_t)nC * (size_t)locap * 2 * sizeof(L2Tmp) > ((size_t)64 << 30)) break; locap *= 2; continue; }
    if (hc[5]) { const size_t need = mm_scaled(c, (size_t)hc[4], sizeof(mm_l2_locus)); c->l2Cap = need + need / 8 + 1024; continue; }
    break;
  }
  if (hc[6] & 1ull) { c->err = "an L1 candidate with more tied L2 loci than 64 GiB of staging can hold"; return MM_ERR_CAPACITY; }
  if (hc[5]) { c->err = "L2 locus buffer overflow"; return MM_ERR_CAPACITY; }
  c->nL2 = (size_t)hc[4];
  return MM_OK;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan; the total stays on the device at *dTotal (a word of dScanTmp, valid until the next scan of this context)
int mm_scan_i32_to_i64_dev(mm_ctx* c, int64_t n, const int32_t* dIn, int64_t* dOut, const int64_t** dTotal) {
  const int64_t nTiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  MM_HIP(c, c->dScanTmp.ensure((size_t)(nTiles + 2) * 8));
  int64_t* tileSum = c->dScanTmp.as<int64_t>();
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)nTiles), dim3(256), 0, c->stream, n, dIn, dOut, tileSum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, c->stream, nTiles, tileSum, tileSum + nTiles);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nTiles), dim3(256), 0, c->stream, n, dOut, tileSum);
  MM_HIP(c, hipGetLastError());
  if (dTotal) *dTotal = tileSum + nTiles;
  return MM_OK;
}
int mm_scan_i32_to_i64(mm_ctx* c, int64_t n, const int32_t* dIn, int64_t* dOut, int64_t* total) {
  const int64_t* dTotal = nullptr;
  const int rc = mm_scan_i32_to_i64_dev(c, n, dIn, dOut, &dTotal);
  if (rc != MM_OK) return rc;
  MM_HIP(c, hipMemcpyAsync(total, dTotal, 8, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  return MM_OK;
}

#define MM_SYNC(c) do { MM_HIP(c, hipStreamSynchronize((c)->stream)); (c)->nSyncs++; } while (0)
#define MM_WIDE_CAP 4096     // steady-state passes: candidates the 16-bit re-run / the exact kernel are launched for without knowing their number
#define MM_EXACT_CAP 1024

int mm_launch_l2(mm_ctx* c, unsigned long long* cnt, bool steady) {
  if (c->windowed || c->P.sketchSize > MM_LDS_MAX_SKETCH) return mm_launch_l2_window(c, cnt);   // fragments longer than segLength (--noSplit), or a sketch no LDS state holds: the literal kernel
  const DeviceIndex& I = c->idx;
  const int s = c->P.sketchSize;
  // sized pass: the candidates are counted (c->nL1) and the candidate-indexed buffers get an eighth of head room, which is what a
  // steady-state pass (count on the device: cnt[2]) launches against
  if (!steady) { const size_t n1 = mm_scaled(c, c->nL1, 96 + 2 * (size_t)((c->P.sketchSize + 3) & ~1)); c->candCap = n1 + n1 / 8 + 1024; }
  const int nC = steady ? (int)c->candCap : (int)c->nL1;              // candidates the launches cover
  const int nCbuf = (int)c->candCap;
  const unsigned long long* nDev = steady ? cnt + 2 : nullptr;
  const int JB = mm_l2_jb(s);                                              // width of the stream entries' sketch-position field
  MM_HIP(c, c->dL2Info.ensure((size_t)nCbuf * sizeof(L2Info) + 64));
  MM_HIP(c, c->dL2Cnt.ensure((size_t)nCbuf * 4 + 64));
  MM_HIP(c, c->dL2Off.ensure((size_t)nCbuf * 8 + 64));
  MM_HIP(c, hipMemsetAsync(cnt + 4, 0, 24, c->stream));
  int64_t totalOps = 0;
  {
    KernelTimer t(c, MM_K_L2_LOCATE);
    hipLaunchKernelGGL(k_l2_extents, dim3((nC + 255) / 256), dim3(256), 0, c->stream, nC, c->P.segLength, JB == 13 ? EF<13>::DELTA_BITS : EF<11>::DELTA_BITS, c->dL1.as<mm_l1_candidate>(), c->dStats.as<mm_frag_stats>(),
                       I.evKey.as<uint32_t>(), I.contigBlock.as<int64_t>(), I.blockOff.as<int64_t>(), I.evBlock.as<int64_t>(),
                       c->dL2Info.as<L2Info>(), c->dL2Cnt.as<int32_t>(), nDev, cnt);
    MM_HIP(c, hipGetLastError());
    if (steady) {
      const int64_t* dTotal = nullptr;
      const int rc = mm_scan_i32_to_i64_dev(c, nC, c->dL2Cnt.as<int32_t>(), c->dL2Off.as<int64_t>(), &dTotal); if (rc != MM_OK) return rc;
      MM_HIP(c, hipMemcpyAsync(c->dCounters.as<unsigned long long>() + 34, dTotal, 8, hipMemcpyDeviceToDevice, c->stream));   // read back with the pass's counters
      hipLaunchKernelGGL(k_l2_gate, dim3(1), dim3(1), 0, c->stream, dTotal, (int64_t)(c->dL2Ops.bytes / 4) - 64, cnt);
      MM_HIP(c, hipGetLastError());
    } else { const int rc = mm_scan_i32_to_i64(c, nC, c->dL2Cnt.as<int32_t>(), c->dL2Off.as<int64_t>(), &totalOps); c->nSyncs++; if (rc != MM_OK) return rc; c->lastOps = (size_t)totalOps; }
  }
  // The located streams (4 bytes per event a candidate touches: ~5 KB per candidate at s = 130) live in HBM only between the locate and
  // the sweep kernels.  A batch with very many candidates -- reads out of repeat families -- is taken through the two in chunks of
  // consecutive candidates whose streams fit MM_L2_STREAM_MIB (default 24 GiB), so the buffer does not grow with the repeat content.
  struct Chunk { int c0, n; int64_t base; };
  std::vector<Chunk> chunks;
  int64_t maxChunkOps = totalOps;
  if (steady) chunks.push_back(Chunk{0, nC, 0});                           // one chunk: the streams must fit the buffer as it is (k_l2_locate flags it otherwise)
  else {
    int64_t budget = (int64_t)24 << 28;                                    // in 4-byte entries: 24 GiB ...
    {
      // ... or half of the device memory that is free (counting what the stream buffer already holds), up to 128 GiB: on a 288 GB part the
      // streams of a configs[4] batch (57 GB: 2.7 M candidates at s = 498) then stay in one chunk, and the passes behind this one are
      // steady-state passes instead of being sized, chunk by chunk, every time
      size_t freeB = 0, totalB = 0;
      if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
        const size_t half = (freeB + c->dL2Ops.bytes) / 2;
        budget = std::max<int64_t>(budget, (int64_t)(std::min<size_t>(half, (size_t)128 << 30) / 4));
      }
    }
    if (const char* e = getenv("MM_L2_STREAM_MIB")) { const double v = atof(e); if (v > 0) budget = (int64_t)(v * 262144.0); }
    if (totalOps <= budget || nC <= 1) chunks.push_back(Chunk{0, nC, 0});
    else {
      const int64_t* dOff = c->dL2Off.as<int64_t>();
      auto offAt = [&](int i, int64_t& v) -> int {                          // opOff[i], opOff[nC] = totalOps
        if (i >= nC) { v = totalOps; return MM_OK; }
        MM_HIP(c, hipMemcpy(&v, dOff + i, 8, hipMemcpyDeviceToHost));
        return MM_OK;
      };
      maxChunkOps = 0;
      int c0 = 0; int64_t base = 0;
      while (c0 < nC) {
        int lo = c0 + 1, hi = nC;                                          // largest c1 with opOff[c1] - base <= budget, at least one candidate
        while (lo < hi) {
          const int mid = lo + (hi - lo + 1) / 2;
          int64_t v; const int rc = offAt(mid, v); if (rc != MM_OK) return rc;
          if (v - base <= budget) lo = mid; else hi = mid - 1;
        }
        int64_t end; { const int rc = offAt(lo, end); if (rc != MM_OK) return rc; }
        chunks.push_back(Chunk{c0, lo - c0, base});
        if (end - base > maxChunkOps) maxChunkOps = end - base;
        c0 = lo; base = end;
      }
      if (getenv("MM_DEBUG")) fprintf(stderr, "[mm] L2: %lld stream entries of %d candidates in %zu chunks of at most %lld\n", (long long)totalOps, nC, chunks.size(), (long long)maxChunkOps);
    }
  }
  if (!steady) {
    const size_t opsFor = chunks.size() == 1 ? mm_scaled(c, (size_t)maxChunkOps, 4) : (size_t)maxChunkOps;
    MM_HIP(c, c->dL2Ops.ensure((opsFor + opsFor / 16) * 4 + 256));   // a sixteenth of head room for the steady-state passes behind this one
    c->l2Chunks = chunks.size();                                                          // a batch that needs several chunks stays with the sized passes
  }
  if (steady && c->dL2Ops.bytes == 0) return MM_PASS_REDO;
  // the pre-load states k_l2_locate leaves for the sweeps: a row of s + 2 cells and four integers per candidate of a chunk
  {
    size_t rows = 0;
    for (const Chunk& ch : chunks) rows = std::max(rows, (size_t)ch.n);
    if (chunks.size() == 1) rows = std::max(rows, (size_t)nCbuf);
    MM_HIP(c, c->dL2InitCells.ensure(rows * (size_t)((s + 3) & ~1) * 2 + 64));
    MM_HIP(c, c->dL2InitState.ensure(rows * sizeof(L2Init) + 64));
  }
  const int initStride = (s + 3) & ~1;                                       // cells per row: s + 2, rounded up to whole dwords
  // buckets of the query-sketch search: at least one per sketch entry (more buckets cost more to fill per candidate than the shorter
  // walks save: profiles/r02z_locate_buckets.txt)
  int NB = 256; while (NB < s) NB <<= 1;
  if (const char* e = getenv("MM_L2_BUCKETS")) { const int v = atoi(e); if (v >= 64 && v <= 16384 && (v & (v - 1)) == 0) NB = v; }
  // waves per workgroup: 4, fewer when four sketches + bucket tables would not fit a CU's LDS
  int wpb = 4; while (wpb > 1 && mm_locate_lds_per_wave(s, NB) * wpb > 160 * 1024) wpb >>= 1;
  const size_t ldsLoc = mm_locate_lds_per_wave(s, NB) * wpb;
  if (ldsLoc > 160 * 1024) { c->err = "sketchSize too large for the LDS-resident query sketch of k_l2_locate"; return MM_ERR_ARG; }
  // total events of the index -> a shift that leaves 16 bits of key (two radix passes)
  int posShift = 0; { const int64_t nEv = (int64_t)(I.evKey.bytes / 4); while ((nEv >> posShift) > 0xFFFF) posShift++; }
  static const bool sortLocate = getenv("MM_L2_LOCATE_NO_SORT") == nullptr;
  // pre-loads of this many records or more go to the exact kernel (12-bit cell counts); MM_L2_PRE_LIMIT lowers it so that tests can send
  // every candidate there
  int preLimit = 4000; if (const char* e = getenv("MM_L2_PRE_LIMIT")) { const int v = atoi(e); if (v >= 0 && v < 4000) preLimit = v; }
  auto locate = [&](const Chunk& ch) -> int {
    KernelTimer t(c, MM_K_L2_LOCATE);
    const int32_t* order = nullptr;
    if (sortLocate && ch.n > 4096) {
      MM_HIP(c, c->dL2OrderPos.ensure((size_t)nCbuf * 4 + 64));
      MM_HIP(c, c->dL2Sort[0].ensure((size_t)nCbuf * 4 + 64)); MM_HIP(c, c->dL2Sort[2].ensure((size_t)nCbuf * 4 + 64));
      hipLaunchKernelGGL(k_l2_pos_keys, dim3((unsigned)((ch.n + 255) / 256)), dim3(256), 0, c->stream, ch.c0, ch.n, posShift, c->dL2Info.as<L2Info>(),
                         c->dL2Sort[0].as<uint32_t>(), c->dL2Sort[2].as<int32_t>(), nDev);
      MM_HIP(c, hipGetLastError());
      const int rc = mm_order_pairs(c, ch.n, 16u, c->dL2OrderPos.as<int32_t>());
      if (rc != MM_OK) return rc;
      order = c->dL2OrderPos.as<int32_t>();
    }
    int blocks = (ch.n + wpb - 1) / wpb; if (blocks > 256 * 32) blocks = 256 * 32;
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsLoc);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(wpb * 64), ldsLoc, c->stream, ch.c0, ch.n, ch.base, s, NB, c->dL1.as<mm_l1_candidate>(),
                         c->dStats.as<mm_frag_stats>(), c->dQHash.as<uint64_t>(), c->dQStrand.as<int8_t>(), c->dSkHash.as<uint64_t>(), c->dSkStrand.as<int8_t>(),
                         I.evKey.as<uint32_t>(),
                         I.evAux.as<uint32_t>(), I.evHash.as<uint64_t>(), I.opKey.as<uint32_t>(), I.opAux.as<uint32_t>(), I.opHash.as<uint64_t>(),
                         I.contigOff.as<int64_t>(),
                         c->dL2Info.as<L2Info>(), c->dL2Off.as<int64_t>(), c->dL2Cnt.as<int32_t>(), c->dL2Ops.as<uint32_t>(), order, cnt, nDev,
                         c->dL2InitCells.as<uint16_t>(), c->dL2InitState.as<L2Init>(), initStride, preLimit);
    };
    if (JB == 13) go(k_l2_locate<13>); else go(k_l2_locate<11>);
    MM_HIP(c, hipGetLastError());
    return MM_OK;
  };
  const bool oneChunk = chunks.size() == 1;
  if (oneChunk) { const int rc = locate(chunks[0]); if (rc != MM_OK) return rc; }      // its streams stay put over the retries below
  // candidates per wave of the sweeps (LPW): 64, fewer when the LDS state of 64 does not fit a CU -- 8-bit cells first, 16-bit cells
  // for the rare candidate whose 5-bit counters overflow
  // (JB == 11: always 64 lanes.  At s = 498 the state of 64 candidates is 32 KB, five waves per CU; 32 lanes per wave double the waves and
  // were measured slower all the same -- sweep 40.1 -> 55.7 ms at configs[4], profiles/NOTES.md round 5: the kernel is issue-bound even there)
  const int lpwN = JB == 11 ? 64 : ((size_t)(s + 2) * 32 <= 160 * 1024 ? 32 : 16);
  const int lpwW = JB == 11 ? ((size_t)(s + 2) * 128 <= 160 * 1024 ? 64 : 32) : ((size_t)(s + 2) * 32 <= 160 * 1024 ? 16 : 8);
  const size_t ldsWide = (size_t)(s + 2) * lpwW * 2;                       // cells 0..S + one of padding, 16 bit
  const size_t ldsNarrow = (((size_t)(s + 2) * lpwN) + 15) & ~(size_t)15;  // 8 bit
  if (ldsWide > 160 * 1024 || ldsNarrow > 160 * 1024) { c->err = "sketchSize too large for the LDS-resident L2 state"; return MM_ERR_ARG; }
  // one launch of a sweep kernel: n candidates starting at c0, or the n listed ones (countDev: their number lives on the device, the
  // launch covers listCap of them)
  auto sweep = [&](bool wide, int c0, int n, int64_t opsBase, const int32_t* list, int locap_, const unsigned long long* countDev, int listCap, int initBase) {
    auto go = [&](auto kern, int lpw, size_t lds) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, dim3((unsigned)((n + lpw - 1) / lpw)), dim3(64), lds, c->stream, c0, n, opsBase, list, c->P.segLength,
                         c->dL1.as<mm_l1_candidate>(), c->dStats.as<mm_frag_stats>(), c->dL2Off.as<int64_t>(), c->dL2Cnt.as<int32_t>(), c->dL2Ops.as<uint32_t>(),
                         c->dL1Off.as<int64_t>(), c->dL2Tmp.as<L2Tmp>(), locap_, c->dL2.as<mm_l2_locus>(), (unsigned long long)c->l2Cap,
                         c->dL2Wide.as<int32_t>(), c->dL2Exact.as<int32_t>(), c->dL2First.as<int64_t>(), c->dL2Num.as<int32_t>(), cnt, countDev, listCap,
                         c->dL2InitCells.as<uint16_t>(), c->dL2InitState.as<L2Init>(), initStride, initBase);
    };
    if (!wide) {
      if (JB == 11) go(k_l2_sweep<false, 11, 64>, 64, ldsNarrow);
      else if (lpwN == 32) go(k_l2_sweep<false, 13, 32>, 32, ldsNarrow);
      else go(k_l2_sweep<false, 13, 16>, 16, ldsNarrow);
    } else {
      if (JB == 11) { if (lpwW == 64) go(k_l2_sweep<true, 11, 64>, 64, ldsWide); else go(k_l2_sweep<true, 11, 32>, 32, ldsWide); }
      else { if (lpwW == 16) go(k_l2_sweep<true, 13, 16>, 16, ldsWide); else go(k_l2_sweep<true, 13, 8>, 8, ldsWide); }
    }
  };
  MM_HIP(c, c->dL2Wide.ensure((size_t)nCbuf * 4 + 64)); MM_HIP(c, c->dL2Exact.ensure((size_t)nCbuf * 4 + 64));
  MM_HIP(c, c->dL2First.ensure((size_t)nCbuf * 8 + 64)); MM_HIP(c, c->dL2Num.ensure((size_t)nCbuf * 4 + 64));
  if (c->l2Cap < (size_t)nCbuf * 2 + 1024) c->l2Cap = (size_t)nCbuf * 2 + 1024;
  unsigned long long hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int locap = steady && c->prevLocap ? c->prevLocap : MM_LOCAP0;
  const bool sortSweep = getenv("MM_L2_NO_SORT") == nullptr;
  for (int attempt = 0; attempt < 24; attempt++) {
    MM_HIP(c, c->dL2.ensure(c->l2Cap * sizeof(mm_l2_locus) + 64));
    MM_HIP(c, c->dL2Tmp.ensure((size_t)nCbuf * locap * sizeof(L2Tmp) + 64));
    MM_HIP(c, hipMemsetAsync(cnt + 4, 0, 16, c->stream));                  // [4] cursor [5] overflow; [6] keeps the locate kernel's flag
    for (const Chunk& ch : chunks) {
      if (!oneChunk) { const int rc = locate(ch); if (rc != MM_OK) return rc; }
      MM_HIP(c, hipMemsetAsync(cnt + 7, 0, 8, c->stream));                 // [7] candidates queued for the wide pass
      MM_HIP(c, hipMemsetAsync(cnt, 0, 8, c->stream));                     // [0] candidates queued for the exact pass (the lookup stage is done with it)
      {
        KernelTimer t(c, MM_K_L2);
        // lane-per-candidate sweep: candidates in order of descending stream length, so that the 64 streams of a wave end together
        // (a wave runs as long as its longest; lengths spread ~ +-10 % around 45 steps: max of 64 is ~15 % above the mean)
        const int32_t* order = nullptr;
        if (sortSweep && ch.n > 64) {
          MM_HIP(c, c->dL2Order.ensure((size_t)nCbuf * 4 + 64));
          const int rc = mm_order_desc(c, c->dL2Cnt.as<int32_t>(), ch.c0, ch.n, 4, c->dL2Order.as<int32_t>());
          if (rc != MM_OK) return rc;
          order = c->dL2Order.as<int32_t>();
        }
        sweep(false, ch.c0, ch.n, ch.base, order, locap, nDev, 0, ch.c0);
        MM_HIP(c, hipGetLastError());
      }
      if (steady) {
        // the 16-bit re-run and the exact kernel for however many candidates the narrow sweep has queued (usually none): launched for
        // a fixed number of them, the lists' lengths are read on the device
        KernelTimer t(c, MM_K_L2);
        sweep(true, 0, MM_WIDE_CAP, ch.base, c->dL2Wide.as<int32_t>(), locap, cnt + 7, MM_WIDE_CAP, ch.c0);
        MM_HIP(c, hipGetLastError());
        MM_HIP(c, c->dL2Cells.ensure((size_t)MM_EXACT_CAP * (size_t)(s + 1) * sizeof(ExactCell) + 64));
        hipLaunchKernelGGL(k_l2_sweep_exact, dim3((unsigned)((MM_EXACT_CAP + 63) / 64)), dim3(64), 0, c->stream, JB, MM_EXACT_CAP, ch.base, c->dL2Exact.as<int32_t>(), c->P.segLength,
                           c->dL1.as<mm_l1_candidate>(), c->dStats.as<mm_frag_stats>(), c->dL2Off.as<int64_t>(), c->dL2Cnt.as<int32_t>(), c->dL2Ops.as<uint32_t>(),
                           c->dL1Off.as<int64_t>(), c->dL2Cells.as<ExactCell>(), s + 1, c->dL2Tmp.as<L2Tmp>(), locap, c->dL2.as<mm_l2_locus>(),
                           (unsigned long long)c->l2Cap, c->dL2First.as<int64_t>(), c->dL2Num.as<int32_t>(), cnt, cnt,
                           c->dL2Info.as<L2Info>(), s, c->dQHash.as<uint64_t>(), c->dQStrand.as<int8_t>(), c->dSkHash.as<uint64_t>(), c->dSkStrand.as<int8_t>(),
                           I.evKey.as<uint32_t>(), I.evAux.as<uint32_t>(), I.evHash.as<uint64_t>(), I.opKey.as<uint32_t>(), I.opAux.as<uint32_t>(), I.opHash.as<uint64_t>());
        MM_HIP(c, hipGetLastError());
        continue;
      }
      MM_HIP(c, hipMemcpyAsync(hc, cnt, 64, hipMemcpyDeviceToHost, c->stream));
      MM_SYNC(c);
      if (hc[7] && !(hc[6] & 1ull) && !hc[5]) {                            // the few candidates whose 5-bit counters overflowed
        const int nWide = (int)hc[7];
        if (getenv("MM_DEBUG")) fprintf(stderr, "[mm] L2 sweep: %d of %d candidates redone with 16-bit cells\n", nWide, ch.n);
        KernelTimer t(c, MM_K_L2);
        sweep(true, 0, nWide, ch.base, c->dL2Wide.as<int32_t>(), locap, nullptr, 0, ch.c0);
        MM_HIP(c, hipGetLastError());
        MM_HIP(c, hipMemcpyAsync(hc, cnt, 64, hipMemcpyDeviceToHost, c->stream));
        MM_SYNC(c);
      }
      if (hc[0] && !(hc[6] & 1ull) && !hc[5]) {                            // candidates with a doubly open query hash: the literal sweep
        const int nExact = (int)hc[0];
        if (getenv("MM_DEBUG")) fprintf(stderr, "[mm] L2 sweep: %d of %d candidates redone by the exact kernel (overlapping windows of one hash)\n", nExact, ch.n);
        MM_HIP(c, c->dL2Cells.ensure((size_t)nExact * (size_t)(s + 1) * sizeof(ExactCell) + 64));
        KernelTimer t(c, MM_K_L2);
        hipLaunchKernelGGL(k_l2_sweep_exact, dim3((unsigned)((nExact + 63) / 64)), dim3(64), 0, c->stream, JB, nExact, ch.base, c->dL2Exact.as<int32_t>(), c->P.segLength,
                           c->dL1.as<mm_l1_candidate>(), c->dStats.as<mm_frag_stats>(), c->dL2Off.as<int64_t>(), c->dL2Cnt.as<int32_t>(), c->dL2Ops.as<uint32_t>(),
                           c->dL1Off.as<int64_t>(), c->dL2Cells.as<ExactCell>(), s + 1, c->dL2Tmp.as<L2Tmp>(), locap, c->dL2.as<mm_l2_locus>(),
                           (unsigned long long)c->l2Cap, c->dL2First.as<int64_t>(), c->dL2Num.as<int32_t>(), cnt, (const unsigned long long*)nullptr,
                           c->dL2Info.as<L2Info>(), s, c->dQHash.as<uint64_t>(), c->dQStrand.as<int8_t>(), c->dSkHash.as<uint64_t>(), c->dSkStrand.as<int8_t>(),
                           I.evKey.as<uint32_t>(), I.evAux.as<uint32_t>(), I.evHash.as<uint64_t>(), I.opKey.as<uint32_t>(), I.opAux.as<uint32_t>(), I.opHash.as<uint64_t>());
        MM_HIP(c, hipGetLastError());
        MM_HIP(c, hipMemcpyAsync(hc, cnt, 64, hipMemcpyDeviceToHost, c->stream));
        MM_SYNC(c);
      }
      if (hc[6] & 1ull) break;                                             // slots ran out: everything is redone below with more of them (a full locus
                                                                           // buffer lets the other chunks run on, so that the cursor ends at the total demand)
    }
    if (steady) return MM_OK;                                              // the flags and the count are read when the pass is over
    if (hc[6] & 1ull) {                                                    // a candidate with more tied loci than slots (tandem repeats): more slots, again
      if ((size_t)nC * (size_t)locap * 2 * sizeof(L2Tmp) > ((size_t)64 << 30)) break;
      locap *= 2;
      const unsigned long long flags = hc[6] & ~1ull;
      MM_HIP(c, hipMemcpyAsync(cnt + 6, &flags, 8, hipMemcpyHostToDevice, c->stream));
      continue;
    }
    if (hc[5]) { const size_t need = mm_scaled(c, (size_t)hc[4], sizeof(mm_l2_locus)); c->l2Cap = need + need / 8 + 1024; continue; }
    break;
  }
  if (hc[6] & 4ull) { c->err = "internal: an L2 stream outgrew the reservation k_l2_extents made for it"; return MM_ERR_CAPACITY; }
  if (hc[6] & 1ull) { c->err = "an L1 candidate with more tied L2 loci than 64 GiB of staging can hold"; return MM_ERR_CAPACITY; }
  if (hc[5]) { c->err = "L2 locus buffer overflow"; return MM_ERR_CAPACITY; }
  c->nL2 = (size_t)hc[4];
  c->prevLocap = locap;
  return MM_OK;
}

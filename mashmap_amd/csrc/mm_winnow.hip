// mashmap_amd/csrc/mm_winnow.hip -- CommonFunc::addMinmers on the device (src/map/include/commonFunc.hpp:302-570):
// the sliding bottom-s "minmer" intervals of one reference contig, from the per-position canonical hashes that
// k_ref_hash (mm_index.hip) leaves in HBM.
//
// The reference slides a window of w - k + 1 k-mer positions one base at a time and keeps the s smallest distinct hashes
// of the window in a std::map (the sketch), everything else in a heap.  A record [wpos, wpos_end) is emitted whenever a hash
// stops being in the sketch or the sign of its strand sum passes through zero.  Two observations make this parallel:
//   * only k-mers whose hash is below a small cut (2.5x the expected s-th smallest of a window, ~6 % of the positions) can ever
//     be in a sketch: the contig is first compacted to that candidate list (k_cand_count / k_cand_write);
//   * the state at a window W0 (sketch members and their strand sums) is a function of that window alone, so the contig is cut
//     into tiles of TW windows that are simulated independently -- one wavefront per tile, cold-started at its first window,
//     stepping only through the windows at which a candidate arrives or departs (k_winnow_tiles).  Runs that were open when
//     the tile started get their true start from the previous tile's open list afterwards (stitching, host).
// The simulation keeps the reference's event order inside a window step -- departure of k-mer W-1 (:376-410), arrival of
// k-mer W+w-k (:412-438), eviction / refill (:440-505) -- so records come out in the reference's emission order, which is what
// makes the final std::sort (:558) reproduce its tie order.
// A tile whose candidate list cannot fill the sketch (fewer than s distinct candidates in some window: N-rich or low-complexity
// sequence) is re-run in DENSE mode, where every valid k-mer position is a candidate.
#include "mm_internal.h"
#include "mm_device.h"
#include <type_traits>
#include "mm_winnow.h"
#include <algorithm>

#define WN_NONE 0xFFFFFFFFFFFFFFFFULL

// ---------------------------------------------------------------------------------------------
// candidate compaction: positions with H <= cap, in position order (block = 256 threads x 8 positions)
// ---------------------------------------------------------------------------------------------
#define CAND_ITEMS 8
#define CAND_TILE (256 * CAND_ITEMS)
__global__ void __launch_bounds__(256)
k_cand_count(const uint64_t* __restrict__ H, int64_t nPos, uint64_t cap, int32_t* __restrict__ blockCnt) {
  __shared__ int sm[4];
  const int64_t base = (int64_t)blockIdx.x * CAND_TILE + (int64_t)threadIdx.x * CAND_ITEMS;
  int c = 0;
#pragma unroll
  for (int i = 0; i < CAND_ITEMS; i++) if (base + i < nPos && H[base + i] <= cap) c++;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blockCnt[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void __launch_bounds__(256)
k_cand_write(const uint64_t* __restrict__ H, const int8_t* __restrict__ ST, int64_t nPos, uint64_t cap, const int64_t* __restrict__ blockOff,
             int32_t* __restrict__ cPos, uint64_t* __restrict__ cHash, int8_t* __restrict__ cSt) {
  __shared__ int sm[256];
  const int64_t base = (int64_t)blockIdx.x * CAND_TILE + (int64_t)threadIdx.x * CAND_ITEMS;
  uint64_t h[CAND_ITEMS]; int c = 0;
#pragma unroll
  for (int i = 0; i < CAND_ITEMS; i++) { h[i] = base + i < nPos ? H[base + i] : WN_NONE; if (h[i] <= cap) c++; }
  sm[threadIdx.x] = c;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int t = (int)threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  int64_t at = blockOff[blockIdx.x] + sm[threadIdx.x] - c;
#pragma unroll
  for (int i = 0; i < CAND_ITEMS; i++)
    if (h[i] <= cap) { cPos[at] = (int32_t)(base + i); cHash[at] = h[i]; cSt[at] = ST[base + i]; at++; }
}

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
// reductions over the 64 lanes on the DPP lanes (mm_device.h): the kernel below is one wave per workgroup with wave-uniform control flow,
// so all lanes are active wherever these are called.  (ds_bpermute shuffles -- an LDS round trip per step, twelve per 64-bit minimum --
// were a fifth of a refill step.)
__device__ __forceinline__ int wn_sum(int v) { return mm_wave_sum(v); }
__device__ __forceinline__ uint64_t wn_min64(uint64_t v) {
  auto step = [&](auto ctrl) {
    constexpr int C = decltype(ctrl)::value;
    const uint64_t y = ((uint64_t)(uint32_t)mm_dpp0<C>((int)(v >> 32)) << 32) | (uint32_t)mm_dpp0<C>((int)(uint32_t)v);   // these controls give every lane a source: no fill
    v = y < v ? y : v;
  };
  step(std::integral_constant<int, MM_DPP_QUAD_1032>()); step(std::integral_constant<int, MM_DPP_QUAD_2301>());
  step(std::integral_constant<int, MM_DPP_ROW_HALF_MIRROR>()); step(std::integral_constant<int, MM_DPP_ROW_MIRROR>());
  uint64_t r = ~0ull;
#pragma unroll
  for (int row = 0; row < 4; row++) {
    const uint64_t x = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), row * 16) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, row * 16);
    r = x < r ? x : r;
  }
  return r;
}

// the sketch of the current window: n entries ascending by hash, in this wave's LDS
struct WnSketch {
  uint64_t* h; int32_t* start; int32_t* sum; int n;
  __device__ __forceinline__ uint64_t maxHash() const { return n > 0 ? h[n - 1] : 0ull; }
  __device__ __forceinline__ int getSum(int p) const { return sum[p]; }
  __device__ __forceinline__ int getStart(int p) const { return start[p]; }
  __device__ __forceinline__ void setStart(int p, int v, int lane) { if (lane == 0) start[p] = v; }
  __device__ __forceinline__ void setSum(int p, int v, int lane) { if (lane == 0) sum[p] = v; __threadfence_block(); }
  __device__ __forceinline__ void popMax(uint64_t& hh, int& st, int& sm, int lane) { hh = h[n - 1]; st = start[n - 1]; sm = sum[n - 1]; n--; }
  __device__ __forceinline__ int find(uint64_t x, int lane) const {           // wave-uniform x; index or -1
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      const uint64_t m = __ballot(i < n && h[i] == x);
      if (m) return base + (int)__builtin_ctzll(m);
    }
    return -1;
  }
  __device__ __forceinline__ bool contains_lane(uint64_t x) const {           // per-lane x: binary search
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (h[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < n && h[lo] == x;
  }
  __device__ __forceinline__ void insert(uint64_t x, int st, int sm, int lane) {
    int p = 0;
    for (int base = 0; base < n; base += 64) { const int i = base + lane; p += __popcll(__ballot(i < n && h[i] < x)); }
    for (int base = ((n > 0 ? n - 1 : 0) / 64) * 64; base >= 0; base -= 64) {  // shift [p, n) up by one, top chunk first
      const int i = base + lane;
      const bool mv = i >= p && i < n;
      uint64_t a = 0; int32_t b = 0, c = 0;
      if (mv) { a = h[i]; b = start[i]; c = sum[i]; }
      __threadfence_block();
      if (mv) { h[i + 1] = a; start[i + 1] = b; sum[i + 1] = c; }
      __threadfence_block();
    }
    if (lane == 0) { h[p] = x; start[p] = st; sum[p] = sm; }
    __threadfence_block();
    n++;
  }
  __device__ __forceinline__ void remove(int p, int lane) {
    for (int base = (p / 64) * 64; base < n; base += 64) {                     // shift (p, n) down by one, bottom chunk first
      const int i = base + lane;
      const bool mv = i > p && i < n;
      uint64_t a = 0; int32_t b = 0, c = 0;
      if (mv) { a = h[i]; b = start[i]; c = sum[i]; }
      __threadfence_block();
      if (mv) { h[i - 1] = a; start[i - 1] = b; sum[i - 1] = c; }
      __threadfence_block();
    }
    n--;
  }
};


// The window's sketch for sizes LDS does not hold (k_winnow_tiles<.., GSK>): the same ordered set, as BLOCKS of at most 64 entries in HBM
// (one entry per lane: a block is loaded, edited in registers and stored by the whole wave) under a directory in LDS (per block, in
// ascending order: its id, its fill and its largest hash).  An insert or a removal touches one block -- 16 bytes per lane there and back
// -- and walks the directory (64 blocks per step), instead of shifting half the sketch through memory as the flat form would (measured:
// 60 s of index build for 3 Mbp at sketchSize 19 998).  A full block splits into two halves; a removal that leaves a block and a neighbour
// with 48 entries or fewer between them merges the two, so that two neighbours always hold more than 48 and the directory stays below
// n / 24 + 2 blocks.  A handle (what find returns) is directory index << 6 | slot; every edit invalidates the handles taken before it.
__device__ __forceinline__ int wn_up1(int v) { return __shfl_up(v, 1); }
__device__ __forceinline__ int wn_down1(int v) { return __shfl_down(v, 1); }
struct WnBlockSketch {
  uint64_t* bh; int32_t* bstart; int32_t* bsum;                   // HBM: block id * 64 + slot
  uint64_t* dMax; uint16_t* dBlk; uint16_t* dCnt; uint16_t* freeList;   // LDS: directory (dirCap entries) + the stack of free block ids
  int n, nBlk, nFree, dirCap;
  __device__ __forceinline__ void init(unsigned char* hbm, unsigned char* lds, int cap, int lane) {
    bh = (uint64_t*)hbm; bstart = (int32_t*)(hbm + (size_t)cap * 64 * 8); bsum = bstart + (size_t)cap * 64;
    dMax = (uint64_t*)lds; dBlk = (uint16_t*)(lds + (size_t)cap * 8); dCnt = dBlk + cap; freeList = dCnt + cap;
    n = 0; nBlk = 0; nFree = cap; dirCap = cap;
    for (int i = lane; i < cap; i += 64) freeList[i] = (uint16_t)(cap - 1 - i);
    __threadfence_block();
  }
  static __host__ __device__ size_t ldsBytes(int cap) { return ((size_t)cap * (8 + 2 + 2 + 2) + 15) & ~(size_t)15; }
  static __host__ __device__ size_t hbmBytes(int cap) { return (size_t)cap * 64 * 16; }
  static __host__ __device__ int capFor(int s) { return s / 8 + 32; }     // (the merge rule keeps the directory near n / 24 blocks; a split next to a small block may break it locally)
  __device__ __forceinline__ uint64_t maxHash() const { return nBlk > 0 ? dMax[nBlk - 1] : 0ull; }
  // first block whose largest hash is >= x, or nBlk
  __device__ __forceinline__ int blockOf(uint64_t x, int lane) const {
    for (int base = 0; base < nBlk; base += 64) {
      const int i = base + lane;
      const uint64_t m = __ballot(i < nBlk && dMax[i] >= x);
      if (m) return base + (int)__builtin_ctzll(m);
    }
    return nBlk;
  }
  __device__ __forceinline__ int find(uint64_t x, int lane) const {
    const int i = blockOf(x, lane);
    if (i >= nBlk) return -1;
    const int b = dBlk[i], c = dCnt[i];
    const uint64_t m = __ballot(lane < c && bh[(size_t)b * 64 + lane] == x);
    return m ? (i << 6) | (int)__builtin_ctzll(m) : -1;
  }
  __device__ __forceinline__ size_t at(int p) const { return (size_t)dBlk[p >> 6] * 64 + (size_t)(p & 63); }
  __device__ __forceinline__ int getSum(int p) const { return bsum[at(p)]; }
  __device__ __forceinline__ int getStart(int p) const { return bstart[at(p)]; }
  __device__ __forceinline__ void setStart(int p, int v, int lane) { if (lane == 0) bstart[at(p)] = v; __threadfence_block(); }
  __device__ __forceinline__ void setSum(int p, int v, int lane) { if (lane == 0) bsum[at(p)] = v; __threadfence_block(); }
  __device__ __forceinline__ void dirInsert(int i, int b, int c, uint64_t mx, int lane) {      // a new directory entry at index i
    for (int base = ((nBlk > 0 ? nBlk - 1 : 0) / 64) * 64; base >= 0; base -= 64) {            // shift [i, nBlk) up by one, top chunk first
      const int j = base + lane;
      const bool mv = j >= i && j < nBlk;
      uint64_t a = 0; uint16_t bb = 0, cc = 0;
      if (mv) { a = dMax[j]; bb = dBlk[j]; cc = dCnt[j]; }
      __threadfence_block();
      if (mv) { dMax[j + 1] = a; dBlk[j + 1] = bb; dCnt[j + 1] = cc; }
      __threadfence_block();
    }
    if (lane == 0) { dMax[i] = mx; dBlk[i] = (uint16_t)b; dCnt[i] = (uint16_t)c; }
    __threadfence_block();
    nBlk++;
  }
  __device__ __forceinline__ void dirRemove(int i, int lane) {
    if (lane == 0) freeList[nFree] = dBlk[i];
    nFree++;
    __threadfence_block();
    for (int base = (i / 64) * 64; base < nBlk; base += 64) {                                  // shift (i, nBlk) down by one, bottom chunk first
      const int j = base + lane;
      const bool mv = j > i && j < nBlk;
      uint64_t a = 0; uint16_t bb = 0, cc = 0;
      if (mv) { a = dMax[j]; bb = dBlk[j]; cc = dCnt[j]; }
      __threadfence_block();
      if (mv) { dMax[j - 1] = a; dBlk[j - 1] = bb; dCnt[j - 1] = cc; }
      __threadfence_block();
    }
    nBlk--;
  }
  __device__ __forceinline__ int takeBlock() { nFree--; return (int)freeList[nFree]; }         // (the directory's capacity bounds the blocks in use: capFor)
  // returns false when the directory is full (cannot happen while the merge rule holds; the tile then fails and is reported)
  __device__ __forceinline__ bool insert(uint64_t x, int st, int sm, int lane) {
    if (nBlk == 0) {
      const int b = takeBlock();
      if (lane == 0) { bh[(size_t)b * 64] = x; bstart[(size_t)b * 64] = st; bsum[(size_t)b * 64] = sm; }
      __threadfence_block();
      dirInsert(0, b, 1, x, lane);
      n++;
      return true;
    }
    int i = blockOf(x, lane);
    if (i >= nBlk) i = nBlk - 1;                                   // larger than everything: the last block takes it
    int b = dBlk[i], c = dCnt[i];
    uint64_t hv = lane < c ? bh[(size_t)b * 64 + lane] : ~0ull; int sv = lane < c ? bstart[(size_t)b * 64 + lane] : 0, mv = lane < c ? bsum[(size_t)b * 64 + lane] : 0;
    if (c == 64) {                                                 // split: the upper half moves into a block of its own behind this one
      if (nBlk >= dirCap || nFree <= 0) return false;
      const int b2 = takeBlock();
      if (lane >= 32) { bh[(size_t)b2 * 64 + lane - 32] = hv; bstart[(size_t)b2 * 64 + lane - 32] = sv; bsum[(size_t)b2 * 64 + lane - 32] = mv; }
      __threadfence_block();
      const uint64_t loMax = ((uint64_t)(uint32_t)__shfl((int)(hv >> 32), 31) << 32) | (uint32_t)__shfl((int)(uint32_t)hv, 31);
      const uint64_t hiMax = ((uint64_t)(uint32_t)__shfl((int)(hv >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)hv, 63);
      if (lane == 0) { dCnt[i] = 32; dMax[i] = loMax; }
      __threadfence_block();
      dirInsert(i + 1, b2, 32, hiMax, lane);
      if (x > loMax) {                                             // x belongs to the new upper block: reload it one entry per lane
        i = i + 1; b = b2;
        hv = lane < 32 ? bh[(size_t)b * 64 + lane] : ~0ull; sv = lane < 32 ? bstart[(size_t)b * 64 + lane] : 0; mv = lane < 32 ? bsum[(size_t)b * 64 + lane] : 0;
      } else if (lane >= 32) { hv = ~0ull; }
      c = 32;
    }
    const int pos = __popcll(__ballot(lane < c && hv < x));        // entries below x
    // lanes [pos, c) move up by one
    const uint64_t hUp = ((uint64_t)(uint32_t)wn_up1((int)(hv >> 32)) << 32) | (uint32_t)wn_up1((int)(uint32_t)hv);
    const int sUp = wn_up1(sv), mUp = wn_up1(mv);
    if (lane == pos) { hv = x; sv = st; mv = sm; }
    else if (lane > pos && lane <= c) { hv = hUp; sv = sUp; mv = mUp; }
    if (lane >= pos && lane <= c) { bh[(size_t)b * 64 + lane] = hv; bstart[(size_t)b * 64 + lane] = sv; bsum[(size_t)b * 64 + lane] = mv; }
    if (lane == 0) { dCnt[i] = (uint16_t)(c + 1); if (pos == c) dMax[i] = x; }
    __threadfence_block();
    n++;
    return true;
  }
  __device__ __forceinline__ void mergeIfSmall(int i, int lane) {                              // blocks i and i + 1
    if (i < 0 || i + 1 >= nBlk) return;
    const int c0 = dCnt[i], c1 = dCnt[i + 1];
    if (c0 + c1 > 48) return;
    const int b0 = dBlk[i], b1 = dBlk[i + 1];
    if (lane < c1) {
      bh[(size_t)b0 * 64 + c0 + lane] = bh[(size_t)b1 * 64 + lane]; bstart[(size_t)b0 * 64 + c0 + lane] = bstart[(size_t)b1 * 64 + lane];
      bsum[(size_t)b0 * 64 + c0 + lane] = bsum[(size_t)b1 * 64 + lane];
    }
    if (lane == 0) { dCnt[i] = (uint16_t)(c0 + c1); dMax[i] = dMax[i + 1]; }
    __threadfence_block();
    dirRemove(i + 1, lane);
  }
  __device__ __forceinline__ void remove(int p, int lane) {
    const int i = p >> 6, sl = p & 63;
    const int b = dBlk[i], c = dCnt[i];
    if (c == 1) { dirRemove(i, lane); n--; mergeIfSmall(i - 1, lane); return; }
    // lanes (sl, c) move down by one
    uint64_t hv = lane < c ? bh[(size_t)b * 64 + lane] : 0ull; int sv = lane < c ? bstart[(size_t)b * 64 + lane] : 0, mv = lane < c ? bsum[(size_t)b * 64 + lane] : 0;
    const uint64_t hDn = ((uint64_t)(uint32_t)wn_down1((int)(hv >> 32)) << 32) | (uint32_t)wn_down1((int)(uint32_t)hv);
    const int sDn = wn_down1(sv), mDn = wn_down1(mv);
    if (lane >= sl && lane < c - 1) { bh[(size_t)b * 64 + lane] = hDn; bstart[(size_t)b * 64 + lane] = sDn; bsum[(size_t)b * 64 + lane] = mDn; }
    // the block's new largest hash: entry c - 2 after the shift = old entry c - 1 unless that was the one removed
    const int srcMax = sl == c - 1 ? c - 2 : c - 1;
    const uint64_t newMax = ((uint64_t)(uint32_t)__shfl((int)(hv >> 32), srcMax) << 32) | (uint32_t)__shfl((int)(uint32_t)hv, srcMax);
    if (lane == 0) { dCnt[i] = (uint16_t)(c - 1); dMax[i] = newMax; }
    __threadfence_block();
    n--;
    mergeIfSmall(i, lane);
    mergeIfSmall(i - 1, lane);
  }
  __device__ __forceinline__ void popMax(uint64_t& hh, int& st, int& sm, int lane) {
    const int p = ((nBlk - 1) << 6) | ((int)dCnt[nBlk - 1] - 1);
    hh = bh[at(p)]; st = bstart[at(p)]; sm = bsum[at(p)];
    remove(p, lane);
  }
};

// ---------------------------------------------------------------------------------------------
// k_winnow_tiles<DENSE>: one wavefront per tile.
//   index space: DENSE -> k-mer positions (H[i] == NONE: not a k-mer);  sparse -> indices into the candidate list
// ---------------------------------------------------------------------------------------------
// GSK: the sketch of the window lives in HBM (skScratch: (s + 1) * 16 bytes per tile slot) instead of LDS -- sketches beyond what a CU's
// 160 KB hold, and from sketchSize 4 097 on (MM_WINNOW_LDS_SKETCH: --dense at segments of 41 kbp and more, or a user's -J); the wave is the only reader and writer of
// its slot, and its accesses are ordered by the same workgroup-scope fences as the LDS form's.
template <bool DENSE, bool GSK = false>
__global__ void __launch_bounds__(64)
k_winnow_tiles(const int32_t* __restrict__ tileList, unsigned char* __restrict__ skScratch,
               const int32_t* __restrict__ cPos, const uint64_t* __restrict__ cHash, const int8_t* __restrict__ cSt, int64_t nCand,
               const uint64_t* __restrict__ H, const int8_t* __restrict__ ST,
               int len, int k, int w, int s, int TW, int nW, int ldsCand,
               mm_minmer* __restrict__ out, int outCap, int32_t* __restrict__ outCount,
               WnOpenRun* __restrict__ open, int32_t* __restrict__ openCount, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int slot = blockIdx.x;                                   // output slot (== position in tileList when there is one)
  const int t = tileList ? tileList[blockIdx.x] : (int)blockIdx.x;
  const int wk = w - k + 1;                                      // k-mer positions per window
  const int W0 = t * TW;
  const int Wend = ((int64_t)(t + 1) * TW < (int64_t)nW - 1) ? (t + 1) * TW : nW - 1;
  const bool lastTile = Wend == nW - 1;

  typename std::conditional<GSK, WnBlockSketch, WnSketch>::type sk;
  unsigned char* stage;
  if constexpr (GSK) {
    const int cap = WnBlockSketch::capFor(s);
    sk.init(skScratch + (size_t)slot * WnBlockSketch::hbmBytes(cap), smem, cap, lane);
    stage = smem + WnBlockSketch::ldsBytes(cap);
  } else {
    sk.h = (uint64_t*)smem;
    sk.start = (int32_t*)(smem + (size_t)(s + 1) * 8);
    sk.sum = sk.start + (s + 1);
    sk.n = 0;
    stage = smem + (size_t)(s + 1) * 16;
  }

  // candidate view (flat pointers: LDS when the tile's candidates were staged, HBM otherwise)
  const int32_t* vPos = cPos; const uint64_t* vHash = DENSE ? H : cHash; const int8_t* vSt = DENSE ? ST : cSt;
  int64_t cLo = 0, cHi = 0, vOff = 0;                            // sparse: candidate indices of positions [W0, Wend + wk - 1]
  if (!DENSE) {
    auto lower = [&](int pos) { int64_t lo = 0, hi = nCand; while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (cPos[mid] < pos) lo = mid + 1; else hi = mid; } return lo; };
    cLo = lower(W0); cHi = lower(Wend + wk);
    if (cHi - cLo <= (int64_t)ldsCand) {
      const int m = (int)(cHi - cLo);
      uint64_t* lH = (uint64_t*)stage; int32_t* lP = (int32_t*)(stage + (size_t)ldsCand * 8); int8_t* lS = (int8_t*)(stage + (size_t)ldsCand * 12);
      for (int i = lane; i < m; i += 64) { lH[i] = cHash[cLo + i]; lP[i] = cPos[cLo + i]; lS[i] = cSt[cLo + i]; }
      __threadfence_block();
      vHash = lH; vPos = lP; vSt = lS; vOff = cLo;      // (no negative-offset pointer: an LDS address must stay inside its 32-bit aperture)
    }
  }
  auto posOf = [&](int64_t i) -> int { return DENSE ? (int)i : vPos[i - vOff]; };
  auto hashAt = [&](int64_t i) -> uint64_t { return vHash[i - vOff]; };
  auto stAt = [&](int64_t i) -> int { return (int)vSt[i - vOff]; };
  auto validAt = [&](int64_t i) -> bool { return DENSE ? hashAt(i) != WN_NONE : true; };

  mm_minmer* myOut = out + (size_t)slot * outCap;
  int nOut = 0; bool fail = false;
  auto emit = [&](uint64_t hh, int st, int en, int sm) {
    if (nOut < outCap) { if (lane == 0) myOut[nOut] = mm_minmer{hh, st, en, 0, (int16_t)sm, 0}; }
    else fail = true;
    nOut++;
  };
  // occurrences of g among the indices [a, b): count and strand sum
  // (both scans below walk the window's candidates 8 x 64 at a time: eight loads in flight per lane before the first compare -- at large
  // sketches the window holds tens of thousands of candidates, and a load per step left the wave waiting for memory 390 times per call)
  auto occ = [&](uint64_t g, int64_t a, int64_t b, int& cnt, int& sm) {
    int c = 0, sgn = 0;
    for (int64_t base = a; base < b; base += 512) {
      uint64_t hv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int64_t i = base + u * 64 + lane; hv[u] = i < b ? hashAt(i) : WN_NONE; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const bool hit = hv[u] == g && base + u * 64 + lane < b;
        c += hit ? 1 : 0;
        if (hit) sgn += stAt(base + u * 64 + lane);
      }
    }
    cnt = wn_sum(c); sm = wn_sum(sgn);
  };
  // smallest hash among [a, b) that is not in the sketch (the reference's heap front)
  auto pendMin = [&](int64_t a, int64_t b) -> uint64_t {
    uint64_t best = WN_NONE;
    const uint64_t mx = sk.maxHash();
    const bool any = sk.n > 0;
    for (int64_t base = a; base < b; base += 512) {
      uint64_t hvs[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int64_t i = base + u * 64 + lane; hvs[u] = i < b ? hashAt(i) : WN_NONE; }   // (DENSE: a position that is no k-mer holds WN_NONE)
#pragma unroll
      for (int u = 0; u < 8; u++) {
        uint64_t hv = hvs[u];
        // The sketch holds the min(s, distinct) smallest hashes of the window -- except for a newcomer that has just arrived below the
        // maximum, and refill() puts that one in before it asks for a minimum -- so a window hash at or below the sketch's maximum IS a
        // member: no search.  MM_WINNOW_CHECK builds verify it with the binary search this line used to be.
#ifdef MM_WINNOW_CHECK
        if constexpr (!GSK) { if (hv != WN_NONE && any && hv <= mx && !sk.contains_lane(hv)) __builtin_trap(); }
#endif
        if (hv != WN_NONE && any && hv <= mx) hv = WN_NONE;
        best = hv < best ? hv : best;
      }
    }
    return wn_min64(best);
  };
  auto skInsert = [&](uint64_t x, int st, int sm) {
    if constexpr (GSK) { if (!sk.insert(x, st, sm, lane)) fail = true; }      // (a full directory: cannot happen while the merge rule holds)
    else sk.insert(x, st, sm, lane);
  };
  auto refill = [&](int startVal, int64_t a, int64_t b, bool haveNew, uint64_t hNew) {   // :487-505
    // a newcomer below the sketch's maximum is the smallest hash outside the sketch (everything else outside is above the maximum,
    // the evicted one included): it goes in first, which restores the invariant pendMin relies on
    if (haveNew && sk.n > 0 && sk.n < s && hNew <= sk.maxHash()) { int cnt, sm; occ(hNew, a, b, cnt, sm); skInsert(hNew, startVal, sm); }
    while (sk.n < s && !fail) {
      const uint64_t pm = pendMin(a, b);
      if (pm == WN_NONE) break;
      int cnt, sm; occ(pm, a, b, cnt, sm);
      skInsert(pm, startVal, sm);
    }
  };

  // window [a, b) in index space
  int64_t a, b;
  if (DENSE) { a = W0; b = (int64_t)W0 + wk; }
  else {
    a = cLo;
    int64_t lo = cLo, hi = cHi; const int lim = W0 + wk;                   // first candidate with pos >= W0 + wk
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (posOf(mid) < lim) lo = mid + 1; else hi = mid; }
    b = lo;
  }
  // cold start: the sketch of window W0; runs that were already open get their start from the previous tile later
  if constexpr (GSK) {
    // refill() asks for the window's smallest hash outside the sketch s times over -- s scans of the window; at these sizes the window's
    // candidates are streamed ONCE instead, in position order, through a bottom-s filter: a member takes the occurrence's strand, a new
    // hash enters while there is room or by pushing the largest member out (whose hash, being at or above every later maximum, cannot
    // come back: its lost sum never matters; a hash turned away stays turned away for the same reason).  Same set, same sums.
    const int startVal = W0 == 0 ? 0 : WN_CARRY;
    for (int64_t base = a; base < b && !fail; base += 64) {
      const int64_t mine = base + lane;
      const uint64_t hL = mine < b ? hashAt(mine) : WN_NONE;
      const int sL = (mine < b && hL != WN_NONE) ? stAt(mine) : 0;
      const int m = (int)((b - base) < 64 ? (b - base) : 64);
      for (int j = 0; j < m && !fail; j++) {
        const uint64_t x = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(hL >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hL, j);
        if (x == WN_NONE) continue;                               // (DENSE: not a k-mer)
        const int st = __builtin_amdgcn_readlane(sL, j);
        const bool below = sk.n > 0 && x <= sk.maxHash();
        const int p = below ? sk.find(x, lane) : -1;
        if (p >= 0) { sk.setSum(p, sk.getSum(p) + st, lane); continue; }
        if (sk.n < s) skInsert(x, startVal, st);
        else if (below) { uint64_t eh; int es, em; sk.popMax(eh, es, em, lane); skInsert(x, startVal, st); }
      }
    }
  } else refill(W0 == 0 ? 0 : WN_CARRY, a, b, false, 0ull);
  if (!DENSE && sk.n < s) fail = true;

  int W = W0;
  const int64_t idxEnd = DENSE ? (int64_t)Wend + wk : cHi;
  while (!fail) {
    int Wn;
    if (DENSE) Wn = W + 1;
    else {
      const int dep = a < b ? posOf(a) + 1 : 0x7fffffff;                     // first candidate of the window leaves at pos + 1
      const int arr = b < idxEnd ? posOf(b) - (wk - 1) : 0x7fffffff;         // next candidate enters when the window end reaches it
      Wn = dep < arr ? dep : arr;
    }
    if (Wn > Wend) break;
    W = Wn;
    bool removed = false, newPending = false; uint64_t hArr = 0;
    // (1) departure of k-mer W-1 (:376-410)
    {
      const bool depValid = DENSE ? validAt(W - 1) : (a < b && posOf(a) == W - 1);
      if (depValid) {
        const int64_t di = DENSE ? (int64_t)(W - 1) : a;
        const uint64_t g = hashAt(di); const int st = stAt(di);
        if (sk.n > 0 && g <= sk.maxHash()) {
          const int p = sk.find(g, lane);
          if (p >= 0) {
            int cnt, sm; occ(g, a, b, cnt, sm);                               // old window: still contains the departing k-mer
            const int cur = sk.getSum(p);
            if (cnt == 1) { emit(g, sk.getStart(p), W, cur); sk.remove(p, lane); removed = true; }
            else {
              if (cur - st == 0 || cur == 0) { emit(g, sk.getStart(p), W, cur); sk.setStart(p, W, lane); }
              sk.setSum(p, cur - st, lane);
            }
          }
        }
      }
      if (DENSE) a = W; else if (depValid) a++;
    }
    // (2) arrival of k-mer W+w-k (:412-438)
    {
      const int ap = W + wk - 1;
      const bool arrValid = DENSE ? validAt(ap) : (b < idxEnd && posOf(b) == ap);
      if (arrValid) {
        const int64_t ai = DENSE ? (int64_t)ap : b;
        const uint64_t hh = hashAt(ai); const int st = stAt(ai);
        const int p = (sk.n > 0 && hh <= sk.maxHash()) ? sk.find(hh, lane) : -1;
        if (p >= 0) {
          const int cur = sk.getSum(p);
          if (cur + st == 0 || cur == 0) { emit(hh, sk.getStart(p), W, cur); sk.setStart(p, W, lane); }
          sk.setSum(p, cur + st, lane);
        } else { newPending = true; hArr = hh; }
      }
      if (DENSE) b = (int64_t)W + wk; else if (arrValid) b++;
    }
    // (3) eviction of the largest member by a smaller newcomer, then refill (:440-505)
    if (sk.n == s && newPending && hArr < sk.maxHash()) {
      uint64_t eh; int es, em;
      sk.popMax(eh, es, em, lane);
      emit(eh, es, W, em);
      removed = true;
    }
    if (sk.n < s && (removed || newPending)) {
      refill(W, a, b, newPending, hArr);
      if (!DENSE && sk.n < s) fail = true;                                  // the cut may be hiding k-mers that belong in the sketch
    }
  }

  if (!fail) {
    if constexpr (GSK) {
      // ascending hash = the directory's blocks in order, a block's entries by lane
      int done = 0;
      for (int i = 0; i < sk.nBlk; i++) {
        const int b = sk.dBlk[i], c = sk.dCnt[i];
        if (lane < c) {
          const uint64_t hh = sk.bh[(size_t)b * 64 + lane]; const int st = sk.bstart[(size_t)b * 64 + lane], sm = sk.bsum[(size_t)b * 64 + lane];
          if (lastTile) { if (nOut + lane < outCap) myOut[nOut + lane] = mm_minmer{hh, st, len - k + 1, 0, (int16_t)sm, 0}; }      // final flush (:509-520)
          else open[(size_t)slot * s + done + lane] = WnOpenRun{hh, st, sm};
        }
        if (lastTile) { if (nOut + c > outCap) fail = true; nOut += c; }
        done += c;
      }
      if (lane == 0) openCount[slot] = lastTile ? 0 : sk.n;
    } else if (lastTile) {                                                   // final flush in ascending hash (:509-520)
      for (int p = 0; p < sk.n; p++) emit(sk.h[p], sk.start[p], len - k + 1, sk.sum[p]);
      if (lane == 0) openCount[slot] = 0;
    } else {
      for (int p = lane; p < sk.n; p += 64) open[(size_t)slot * s + p] = WnOpenRun{sk.h[p], sk.start[p], sk.sum[p]};
      if (lane == 0) openCount[slot] = sk.n;
    }
  }
  if (lane == 0) { outCount[slot] = fail ? 0 : nOut; status[slot] = fail ? 1 : 0; }
}

// gathers the tiles' records into one dense array (block per tile)
__global__ void __launch_bounds__(256)
k_winnow_compact(const mm_minmer* __restrict__ out, int outCap, const int32_t* __restrict__ outCount, const int64_t* __restrict__ outOff,
                 mm_minmer* __restrict__ dense) {
  const int t = blockIdx.x;
  const int n = outCount[t];
  const mm_minmer* src = out + (size_t)t * outCap;
  mm_minmer* dst = dense + outOff[t];
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// host orchestration for one contig whose hashes are resident (dH, dS)
// ---------------------------------------------------------------------------------------------
int mm_scan_i32_to_i64(mm_ctx* c, int64_t n, const int32_t* dIn, int64_t* dOut, int64_t* total);   // mm_l2.hip

static const void* kSparseAttr(bool gsk) { return gsk ? (const void*)k_winnow_tiles<false, true> : (const void*)k_winnow_tiles<false, false>; }
static const void* kDenseAttr(bool gsk) { return gsk ? (const void*)k_winnow_tiles<true, true> : (const void*)k_winnow_tiles<true, false>; }

int mm_winnow_contig_device(mm_ctx* c, WinnowBuffers& B, const uint64_t* dH, const int8_t* dS, int64_t nPos, int len,
                            std::vector<mm_minmer>& records, std::vector<int32_t>& tileCount, std::vector<WnOpenRun>& openRuns,
                            std::vector<int32_t>& openCount, WnStaged* staged) {
  const int k = c->P.kmerSize, w = c->P.segLength, s = c->P.sketchSize;
  const int nW = len - w + 1;
  const int wk = w - k + 1;
  // windows per tile: a segment length's worth -- except with the sketch in HBM (sketchSize > MM_WINNOW_LDS_SKETCH), where a window step costs scans
  // of tens of thousands of candidates and a cold start is one streamed pass: there the tiles shrink (down to w / 16) until there are
  // about four per CU, so that a small reference still fills the GPU
  const bool gskTiles = s > MM_WINNOW_LDS_SKETCH || getenv("MM_WINNOW_GSK") != nullptr;
  const int TW = !gskTiles ? w : (int)std::max<int64_t>((int64_t)std::max(1, w / 16), std::min<int64_t>((int64_t)w, ((int64_t)nW + 1023) / 1024));
  const int nTiles = nW - 1 <= 0 ? 1 : (int)(((int64_t)nW - 1 + TW - 1) / TW);
  // cut: the s-th smallest of wk canonical hashes (min of two uniforms) is ~ s / (2 wk) * 2^64; keep 2.5x that
  uint64_t cap = ~0ull;
  { const double frac = 2.5 * (double)s / (2.0 * (double)wk); if (frac < 0.45) cap = (uint64_t)(frac * 18446744073709551615.0); }
  const bool sparse = cap != ~0ull;
  // ---- candidates
  int64_t nCand = 0;
  if (sparse) {
    const int64_t nBlocks = (nPos + CAND_TILE - 1) / CAND_TILE;
    MM_HIP(c, B.blockCnt.ensure((size_t)nBlocks * 4 + 64)); MM_HIP(c, B.blockOff.ensure((size_t)nBlocks * 8 + 64));
    hipLaunchKernelGGL(k_cand_count, dim3((unsigned)nBlocks), dim3(256), 0, c->stream, dH, nPos, cap, B.blockCnt.as<int32_t>());
    MM_HIP(c, hipGetLastError());
    int rc = mm_scan_i32_to_i64(c, nBlocks, B.blockCnt.as<int32_t>(), B.blockOff.as<int64_t>(), &nCand);
    if (rc != MM_OK) return rc;
    MM_HIP(c, B.cPos.ensure((size_t)nCand * 4 + 64)); MM_HIP(c, B.cHash.ensure((size_t)nCand * 8 + 64)); MM_HIP(c, B.cSt.ensure((size_t)nCand + 64));
    hipLaunchKernelGGL(k_cand_write, dim3((unsigned)nBlocks), dim3(256), 0, c->stream, dH, dS, nPos, cap, B.blockOff.as<int64_t>(),
                       B.cPos.as<int32_t>(), B.cHash.as<uint64_t>(), B.cSt.as<int8_t>());
    MM_HIP(c, hipGetLastError());
  }
  // ---- tiles
  const int outCap = std::max(1024, 8 * s);
  MM_HIP(c, B.out.ensure((size_t)nTiles * outCap * sizeof(mm_minmer) + 64));
  MM_HIP(c, B.outCount.ensure((size_t)nTiles * 4 + 64)); MM_HIP(c, B.openCount.ensure((size_t)nTiles * 4 + 64));
  MM_HIP(c, B.status.ensure((size_t)nTiles * 4 + 64)); MM_HIP(c, B.open.ensure((size_t)nTiles * s * sizeof(WnOpenRun) + 64));
  MM_HIP(c, B.outOff.ensure((size_t)nTiles * 8 + 64));
  // the window's sketch in LDS while it fits next to at least 64 staged candidates (sketchSize <= MM_WINNOW_LDS_SKETCH), in HBM beyond
  const bool gsk = s > MM_WINNOW_LDS_SKETCH || getenv("MM_WINNOW_GSK") != nullptr;   // (MM_WINNOW_GSK=1: the HBM form at any size, for the tests)
  const size_t ldsSketch = gsk ? WnBlockSketch::ldsBytes(WnBlockSketch::capFor(s)) : (size_t)(s + 1) * 16;
  int ldsCand = 1024; while ((size_t)ldsCand * 13 + ldsSketch > 60 * 1024 && ldsCand > 64) ldsCand >>= 1;
  const size_t ldsSparse = ldsSketch + (size_t)ldsCand * 13 + 16;
  const size_t skBytes = WnBlockSketch::hbmBytes(WnBlockSketch::capFor(s));
  if (gsk) MM_HIP(c, B.skScratch.ensure((size_t)nTiles * skBytes + 64));
  unsigned char* skS = gsk ? B.skScratch.as<unsigned char>() : (unsigned char*)nullptr;
  auto kSparse = gsk ? k_winnow_tiles<false, true> : k_winnow_tiles<false, false>;
  auto kDense = gsk ? k_winnow_tiles<true, true> : k_winnow_tiles<true, false>;
  MM_HIP(c, hipFuncSetAttribute((const void*)kSparseAttr(gsk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsSparse));
  MM_HIP(c, hipFuncSetAttribute((const void*)kDenseAttr(gsk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsSketch + 16));
  {
    KernelTimer t(c, MM_K_WINNOW);
    if (sparse)
      hipLaunchKernelGGL(kSparse, dim3(nTiles), dim3(64), ldsSparse, c->stream, (const int32_t*)nullptr, skS,
                         B.cPos.as<int32_t>(), B.cHash.as<uint64_t>(), B.cSt.as<int8_t>(), nCand, dH, dS, len, k, w, s, TW, nW, ldsCand,
                         B.out.as<mm_minmer>(), outCap, B.outCount.as<int32_t>(), B.open.as<WnOpenRun>(), B.openCount.as<int32_t>(), B.status.as<int32_t>());
    else
      hipLaunchKernelGGL(kDense, dim3(nTiles), dim3(64), ldsSketch + 16, c->stream, (const int32_t*)nullptr, skS,
                         (const int32_t*)nullptr, (const uint64_t*)nullptr, (const int8_t*)nullptr, (int64_t)0, dH, dS, len, k, w, s, TW, nW, 0,
                         B.out.as<mm_minmer>(), outCap, B.outCount.as<int32_t>(), B.open.as<WnOpenRun>(), B.openCount.as<int32_t>(), B.status.as<int32_t>());
    MM_HIP(c, hipGetLastError());
  }
  std::vector<int32_t> status((size_t)nTiles);
  MM_HIP(c, hipMemcpyAsync(status.data(), B.status.p, (size_t)nTiles * 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  std::vector<int32_t> redo;
  for (int t = 0; t < nTiles; t++) if (status[t]) redo.push_back(t);
  int64_t total = 0;
  { const int rc = mm_scan_i32_to_i64(c, nTiles, B.outCount.as<int32_t>(), B.outOff.as<int64_t>(), &total); if (rc != MM_OK) return rc; }
  MM_HIP(c, B.dense.ensure((size_t)total * sizeof(mm_minmer) + 64));
  hipLaunchKernelGGL(k_winnow_compact, dim3(nTiles), dim3(256), 0, c->stream, B.out.as<mm_minmer>(), outCap, B.outCount.as<int32_t>(),
                     B.outOff.as<int64_t>(), B.dense.as<mm_minmer>());
  MM_HIP(c, hipGetLastError());
  std::vector<mm_minmer> first;
  tileCount.assign((size_t)nTiles, 0); openCount.assign((size_t)nTiles, 0);
  const size_t recBytes = (size_t)total * sizeof(mm_minmer), runBytes = (size_t)nTiles * s * sizeof(WnOpenRun), recPad = (recBytes + 255) & ~(size_t)255;
  unsigned char* hs = (unsigned char*)B.host(recPad + runBytes + 64);
  if (hs) {
    if (total) MM_HIP(c, hipMemcpyAsync(hs, B.dense.p, recBytes, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(hs + recPad, B.open.p, runBytes, hipMemcpyDeviceToHost, c->stream));
  } else {
    first.resize((size_t)total); openRuns.assign((size_t)nTiles * s, WnOpenRun{0, 0, 0});
    if (total) MM_HIP(c, hipMemcpyAsync(first.data(), B.dense.p, recBytes, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(openRuns.data(), B.open.p, runBytes, hipMemcpyDeviceToHost, c->stream));
  }
  MM_HIP(c, hipMemcpyAsync(tileCount.data(), B.outCount.p, (size_t)nTiles * 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(openCount.data(), B.openCount.p, (size_t)nTiles * 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  if (hs && staged && redo.empty()) {             // the common case: the caller's finishing thread copies the two arrays out
    staged->hs = hs; staged->total = (size_t)total; staged->recPad = recPad; staged->nRuns = (size_t)nTiles * s;
    records.clear(); openRuns.clear();
    return MM_OK;
  }
  if (hs) {
    const mm_minmer* r0 = (const mm_minmer*)hs; const WnOpenRun* o0 = (const WnOpenRun*)(hs + recPad);
    first.assign(r0, r0 + (size_t)total);
    openRuns.assign(o0, o0 + (size_t)nTiles * s);
  }
  if (redo.empty()) { records.swap(first); return MM_OK; }

  // ---- failed tiles again, every valid k-mer a candidate, room for the worst case (3 records per window step + flush)
  const int nRedo = (int)redo.size();
  const int bigCap = 3 * TW + s + 16;
  MM_HIP(c, B.redoList.ensure((size_t)nRedo * 4 + 64));
  MM_HIP(c, B.out2.ensure((size_t)nRedo * bigCap * sizeof(mm_minmer) + 64));
  MM_HIP(c, B.outCount2.ensure((size_t)nRedo * 4 + 64)); MM_HIP(c, B.openCount2.ensure((size_t)nRedo * 4 + 64));
  MM_HIP(c, B.status2.ensure((size_t)nRedo * 4 + 64)); MM_HIP(c, B.open2.ensure((size_t)nRedo * s * sizeof(WnOpenRun) + 64));
  MM_HIP(c, hipMemcpyAsync(B.redoList.p, redo.data(), (size_t)nRedo * 4, hipMemcpyHostToDevice, c->stream));
  {
    KernelTimer t(c, MM_K_WINNOW);
    if (gsk) MM_HIP(c, B.skScratch.ensure((size_t)nRedo * skBytes + 64));
    hipLaunchKernelGGL(kDense, dim3(nRedo), dim3(64), ldsSketch + 16, c->stream, B.redoList.as<int32_t>(),
                       gsk ? B.skScratch.as<unsigned char>() : (unsigned char*)nullptr, (const int32_t*)nullptr, (const uint64_t*)nullptr, (const int8_t*)nullptr, (int64_t)0, dH, dS, len, k, w, s, TW, nW, 0,
                       B.out2.as<mm_minmer>(), bigCap, B.outCount2.as<int32_t>(), B.open2.as<WnOpenRun>(), B.openCount2.as<int32_t>(), B.status2.as<int32_t>());
    MM_HIP(c, hipGetLastError());
  }
  std::vector<int32_t> cnt2((size_t)nRedo), oc2((size_t)nRedo), st2((size_t)nRedo);
  std::vector<WnOpenRun> or2((size_t)nRedo * s);
  std::vector<mm_minmer> rec2((size_t)nRedo * bigCap);
  MM_HIP(c, hipMemcpyAsync(cnt2.data(), B.outCount2.p, (size_t)nRedo * 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(oc2.data(), B.openCount2.p, (size_t)nRedo * 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(st2.data(), B.status2.p, (size_t)nRedo * 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(or2.data(), B.open2.p, (size_t)nRedo * s * sizeof(WnOpenRun), hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(rec2.data(), B.out2.p, (size_t)nRedo * bigCap * sizeof(mm_minmer), hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  for (int r = 0; r < nRedo; r++) if (st2[r]) { c->err = "mm_index_build: winnowing tile overflowed its worst-case record buffer"; return MM_ERR_CAPACITY; }
  // splice: tile order, redone tiles taken from the second launch
  std::vector<int64_t> off((size_t)nTiles + 1, 0);
  for (int t = 0; t < nTiles; t++) off[t + 1] = off[t] + tileCount[t];
  records.clear();
  size_t r = 0;
  for (int t = 0; t < nTiles; t++) {
    if (r < redo.size() && redo[r] == t) {
      records.insert(records.end(), rec2.begin() + (size_t)r * bigCap, rec2.begin() + (size_t)r * bigCap + cnt2[r]);
      tileCount[t] = cnt2[r]; openCount[t] = oc2[r];
      std::copy(or2.begin() + (size_t)r * s, or2.begin() + (size_t)r * s + oc2[r], openRuns.begin() + (size_t)t * s);
      r++;
    } else {
      records.insert(records.end(), first.begin() + off[t], first.begin() + off[t + 1]);
    }
  }
  return MM_OK;
}

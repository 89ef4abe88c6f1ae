// mashmap_amd/csrc/mm_map.hip -- device index + L1/L2 mapping kernels (gfx950).
//
//   mm_build_device_index   flattening of skch::Sketch for the device     winSketch.hpp:100-102
//   k_lookup_l1             getSeedHits (freq. seed removal) + getSeedIntervalPoints + computeL1CandidateRegions, fused:
//                           wave per fragment, points sorted in registers   computeMap.hpp:818-843, 857-912, 916-1116
//   k_sort_points_* / k_l1_sweep   the same through HBM for the fragments the fused path hands over (> 128 points,
//                           -Y groups, position groups spanning contigs) or when MM_OPT_KEEP_POINTS is set
//   (L2 lives in mm_l2.hip)
#include "mm_internal.h"
#include "mm_device.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#define MM_EMPTY MM_HT_EMPTY

// ---------------------------------------------------------------------------------------------
// host: flatten the reference index
// ---------------------------------------------------------------------------------------------
static inline uint64_t pack_point(int32_t seqId, int32_t pos, int side) {
  return ((uint64_t)(uint32_t)seqId << 33) | ((uint64_t)(uint32_t)pos << 1) | (side == 1 ? 1ull : 0ull);
}

// The index as the reference holds it (host mirrors: minmerIndex after dropFreqSeedSet, the lookup map flattened, the frequent seeds)
// -> device arrays -> mm_flatten_device_index (mm_index_dev.hip), which builds everything Map reads on the GPU.
int mm_build_device_index(mm_ctx* c, const int32_t* contigLen, const int32_t* refGroup, size_t nContigs) {
  DeviceIndex& I = c->idx;
  I.ready = false;
  const size_t n = c->hMinmers.size(), nk = c->hKeys.size(), np = c->hPoints.size();
  MM_HIP(c, c->dCounters.ensure(256));
  std::vector<uint64_t> pk(np);
  for (size_t i = 0; i < np; i++) pk[i] = pack_point(c->hPoints[i].seqId, c->hPoints[i].pos, c->hPoints[i].side);
  std::vector<uint8_t> fq(nk, 0);
  for (size_t i = 0; i < nk; i++) fq[i] = std::binary_search(c->hFreq.begin(), c->hFreq.end(), c->hKeys[i]) ? 1 : 0;
  DevBuf dRec;
  MM_HIP(c, dRec.ensure(n * sizeof(mm_minmer) + 64));
  MM_HIP(c, I.keys.ensure(nk * 8 + 64)); MM_HIP(c, I.keyOff.ensure((nk + 1) * 8 + 64)); MM_HIP(c, I.keyFreq.ensure(nk + 64)); MM_HIP(c, I.ptKeys.ensure(np * 8 + 64));
  if (n) MM_HIP(c, hipMemcpyAsync(dRec.p, c->hMinmers.data(), n * sizeof(mm_minmer), hipMemcpyHostToDevice, c->stream));
  if (nk) { MM_HIP(c, hipMemcpyAsync(I.keys.p, c->hKeys.data(), nk * 8, hipMemcpyHostToDevice, c->stream));
            MM_HIP(c, hipMemcpyAsync(I.keyFreq.p, fq.data(), nk, hipMemcpyHostToDevice, c->stream)); }
  MM_HIP(c, hipMemcpyAsync(I.keyOff.p, c->hOffsets.data(), (nk + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (np) MM_HIP(c, hipMemcpyAsync(I.ptKeys.p, pk.data(), np * 8, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  const int rc = mm_flatten_device_index(c, dRec.as<mm_minmer>(), n, nk, np, contigLen, refGroup, nContigs);
  dRec.release();
  c->mirrorMinmers = c->mirrorMap = true;
  return rc;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
// value of lane (lane ^ M), M a power of two, without the LDS crossbar: DPP inside a row of 16, v_permlane{16,32}_swap
// (gfx950) across rows.  All 64 lanes must be active.
template <int M>
__device__ __forceinline__ uint32_t mm_lane_xor32(uint32_t v, int lane) {
  if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, MM_DPP_QUAD_1032, 0xf, 0xf, false);
  else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, MM_DPP_QUAD_2301, 0xf, 0xf, false);
  else if constexpr (M == 4) {                                   // banks 0,2 (lane bit 2 clear) read lane+4, banks 1,3 read lane-4
    const int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104 /* row_shl:4 */, 0xf, 0x5, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xa, false);
  } else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
  else if constexpr (M == 16) {                                  // first' = (r0,r0,r2,r2), second' = (r1,r1,r3,r3)
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (lane & 16) ? r[0] : r[1];
  } else {
    static_assert(M == 32, "lane distance");
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // first' = (lo,lo), second' = (hi,hi)
    return (lane & 32) ? r[0] : r[1];
  }
}
template <int M>
__device__ __forceinline__ uint64_t mm_lane_xor64(uint64_t v, int lane) {
  return ((uint64_t)mm_lane_xor32<M>((uint32_t)(v >> 32), lane) << 32) | mm_lane_xor32<M>((uint32_t)v, lane);
}
__device__ __forceinline__ uint64_t mm_shfl_xor64(uint64_t v, int m, int lane) {   // m: a power of two known after unrolling
  switch (m) {
    case 1: return mm_lane_xor64<1>(v, lane);
    case 2: return mm_lane_xor64<2>(v, lane);
    case 4: return mm_lane_xor64<4>(v, lane);
    case 8: return mm_lane_xor64<8>(v, lane);
    case 16: return mm_lane_xor64<16>(v, lane);
    default: return mm_lane_xor64<32>(v, lane);
  }
}
// neighbour lanes (lane 0 / lane 63 keep their own value, as __shfl_up / __shfl_down do) through DPP wave shifts
__device__ __forceinline__ int mm_shfl_up1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ int mm_shfl_down1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ uint64_t mm_shfl_up64(uint64_t v, int d) {      // d == 1
  return ((uint64_t)(uint32_t)mm_shfl_up1((int)(v >> 32)) << 32) | (uint32_t)mm_shfl_up1((int)(uint32_t)v);
}
__device__ __forceinline__ uint64_t mm_shfl_down64(uint64_t v, int d) {    // d == 1
  return ((uint64_t)(uint32_t)mm_shfl_down1((int)(v >> 32)) << 32) | (uint32_t)mm_shfl_down1((int)(uint32_t)v);
}

struct MapFlags { int hg, skipSelf, skipPrefix, lowerTri; };

// ascending bitonic sort of 64*R keys held R per lane (R a power of two), element index = lane*R + r
template <int R>
__device__ __forceinline__ void mm_wave_bitonic(uint64_t (&k)[R], int lane) {
  constexpr int N = 64 * R;
#pragma unroll
  for (int k2 = 2; k2 <= N; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      // one compare per exchange: the element keeps its key or takes the partner's (equal keys: either)
      if (j < R) {                                   // partner inside the lane
#pragma unroll
        for (int r = 0; r < R; r++) {
          if (r & j) continue;
          const bool up = (((lane * R + r) & k2) == 0);
          const uint64_t a = k[r], b = k[r | j];
          const bool swap = (a < b) != up;
          k[r] = swap ? b : a; k[r | j] = swap ? a : b;
        }
      } else {                                       // partner in lane ^ (j / R), same register
        const int lj = j / R;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int idx = lane * R + r;
          const uint64_t other = mm_shfl_xor64(k[r], lj, lane);
          const bool keepMin = ((idx & k2) == 0) == ((idx & j) == 0);
          const bool lt = k[r] < other;
          k[r] = (lt != keepMin) ? other : k[r];
        }
      }
    }
  }
}

// per-wave LDS scratch of the fused kernel
#define MM_FUSE_MAXRUNS 64
struct L1Run { int32_t seq, start, end, isize; };
template <int MAXPTS>
struct FuseScratchT {
  uint64_t a[MAXPTS];                // gathered points, later (seqId<<32 | pos) of every position group
  int32_t v[MAXPTS];                 // overlap count after every position group
  L1Run run[MM_FUSE_MAXRUNS];        // candidate runs before joining (a fragment with more goes to the literal sweep)
};

// Interval points of the fragment's surviving seeds -> dst[0..P) (skip_self / skip_prefix / lower_triangular applied,
// computeMap.hpp:891-896; dropped points become MM_EMPTY and sort to the end).  Returns the wave-wide count of kept points.
// ids (may be null): the seed every point came from, as a small number unique inside the fragment (its index in the raw sketch) -- what the
// windowLen != 0 sweep counts open windows per hash by (computeMap.hpp:950: hash_to_freq)
// done: points already at dst (updated); rdBase: number of the first round (a sketch probed in several batches calls this per batch)
template <class Dst, class ValAt>
__device__ __forceinline__ int mm_gather_points(Dst dst, int& done, int rdBase, int nRounds, ValAt&& valAt,
                                                const uint64_t* __restrict__ ptKeys, const int32_t* __restrict__ refGroup,
                                                int rg, int self, int seqCounter, MapFlags fl, int lane, uint16_t* __restrict__ ids = nullptr) {
  int nValid = 0;
  auto one = [&](uint64_t key, int at, int seed) {
    const int seqId = (int)(key >> 33);
    bool drop = false;
    if (fl.skipSelf && seqId == self) drop = true;
    if (fl.skipPrefix && refGroup[seqId] == rg) drop = true;
    if (fl.lowerTri && !(seqCounter > seqId)) drop = true;
    if (drop) key = MM_EMPTY; else nValid++;
    dst[at] = key;
    if (ids) ids[at] = (uint16_t)seed;
  };
  for (int rd = 0; rd < nRounds; rd++) {
    const uint64_t val = valAt(rd);                // table value of this lane's seed in round rd (0: none)
    const int c = (int)((val >> 1) & 0x7fffffull);
    const int my = done + mm_wave_excl_scan(c);
    const uint64_t src = val >> 24;
    // a short run is copied by the lane that owns the seed; a long one -- a seed of a repeat family brings hundreds of points, and the
    // other 63 lanes used to wait for the one that copied them point by point -- by the whole wave, 64 consecutive points at a time
    const bool longRun = c > 8;
    if (!longRun) for (int j = 0; j < c; j++) one(ptKeys[src + j], my + j, (rdBase + rd) * 64 + lane);
    uint64_t mLong = mm_ballot(longRun);
    while (mLong) {
      const int l = (int)__builtin_ctzll(mLong); mLong &= mLong - 1ull;
      const int cL = __shfl(c, l), myL = __shfl(my, l);
      const uint64_t srcL = ((uint64_t)(uint32_t)__shfl((int)(src >> 32), l) << 32) | (uint32_t)__shfl((int)(uint32_t)src, l);
      for (int j = lane; j < cL; j += 64) one(ptKeys[srcL + j], myL + j, (rdBase + rd) * 64 + l);
    }
    done += mm_wave_sum(c);
  }
  return mm_wave_sum(nValid);
}

// computeL1CandidateRegions (computeMap.hpp:916-1116, windowLen == 0) on a wave-sorted point list, data-parallel form
// (SURVEY App. A.4): position groups = runs of equal pos; overlap after group g = inclusive prefix sum of (+1 OPEN, -1 CLOSE);
// pass 1 best = max; pass 2 runs of consecutive groups -- the last one excluded -- with overlap >= minimumHits, cut on seqId
// change; runs closer than segLength joined.  Returns -1 when the list needs the literal sweep instead (a position group that
// spans two contigs: there the reference's trailing pointer, which compares (seqId,pos), lags its leading pointer, which
// compares pos only); otherwise the number of candidates, which have been stored at sc.run[0..n).
template <int R, class FuseScratch>
__device__ __forceinline__ int mm_l1_fused(uint64_t (&k)[R], FuseScratch& sc, int sketchSizeQ, int minHits, int hg,
                                           const int32_t* __restrict__ cutoffs, int nCutoffs, int sParam, int segLength, int lane) {
  bool valid[R]; uint64_t prv[R], nxt[R];
#pragma unroll
  for (int e = 0; e < R; e++) valid[e] = k[e] != MM_EMPTY;
  {
    const uint64_t up = mm_shfl_up64(k[R - 1], 1), dn = mm_shfl_down64(k[0], 1);
    prv[0] = lane == 0 ? MM_EMPTY : up;
    nxt[R - 1] = lane == 63 ? MM_EMPTY : dn;
#pragma unroll
    for (int e = 1; e < R; e++) { prv[e] = k[e - 1]; nxt[e - 1] = k[e]; }
  }
  bool mixed = false; int delta[R]; bool gLast[R];
#pragma unroll
  for (int e = 0; e < R; e++) {
    const bool pv = prv[e] != MM_EMPTY, nv = nxt[e] != MM_EMPTY;
    if (valid[e] && pv && (uint32_t)(prv[e] >> 1) == (uint32_t)(k[e] >> 1) && (prv[e] >> 33) != (k[e] >> 33)) mixed = true;
    delta[e] = valid[e] ? ((k[e] & 1ull) ? 1 : -1) : 0;
    gLast[e] = valid[e] && (!nv || (uint32_t)(nxt[e] >> 1) != (uint32_t)(k[e] >> 1));
  }
  if (__ballot(mixed)) return -1;
  int laneSum = 0, laneGroups = 0;
#pragma unroll
  for (int e = 0; e < R; e++) { laneSum += delta[e]; laneGroups += gLast[e] ? 1 : 0; }
  int run = mm_wave_excl_scan(laneSum);
  int gidx = mm_wave_excl_scan(laneGroups);
  const int G = mm_wave_sum(laneGroups);
  int best = 0;
#pragma unroll
  for (int e = 0; e < R; e++) {
    run += delta[e];
    if (gLast[e]) { sc.a[gidx] = k[e] >> 1; sc.v[gidx] = run; gidx++; best = run > best ? run : best; }
  }
  best = mm_wave_max(best);
  if (G == 0) return 0;
  if (hg) {                                                       // computeMap.hpp:984-998
    if (best < minHits) return 0;
    const double div = (double)sParam / 1000.0 > 1.0 ? (double)sParam / 1000.0 : 1.0;
    int ci = (int)((double)(best < sketchSizeQ ? best : sketchSizeQ) / div);
    if (ci >= nCutoffs) ci = nCutoffs - 1;
    const int cut = cutoffs[ci];
    minHits = cut > minHits ? cut : minHits;
  }
  __threadfence_block();
  // pass 2 over the groups, again R per lane
  uint64_t gk[R]; bool flag[R];
#pragma unroll
  for (int e = 0; e < R; e++) {
    const int j = lane * R + e;
    gk[e] = j < G ? sc.a[j] : 0ull;
    flag[e] = j < G - 1 && sc.v[j] >= minHits;
  }
  bool pflag[R], nflag[R]; uint32_t pseq[R], nseq[R];
  {
    const int upf = mm_shfl_up1((int)flag[R - 1]), dnf = mm_shfl_down1((int)flag[0]);
    const int ups = mm_shfl_up1((int)(gk[R - 1] >> 32)), dns = mm_shfl_down1((int)(gk[0] >> 32));
    pflag[0] = lane != 0 && upf; pseq[0] = (uint32_t)ups;
    nflag[R - 1] = lane != 63 && dnf; nseq[R - 1] = (uint32_t)dns;
#pragma unroll
    for (int e = 1; e < R; e++) { pflag[e] = flag[e - 1]; pseq[e] = (uint32_t)(gk[e - 1] >> 32); nflag[e - 1] = flag[e]; nseq[e - 1] = (uint32_t)(gk[e] >> 32); }
  }
  bool rStart[R], rEnd[R]; int laneStarts = 0;
#pragma unroll
  for (int e = 0; e < R; e++) {
    const uint32_t sq = (uint32_t)(gk[e] >> 32);
    rStart[e] = flag[e] && (!pflag[e] || pseq[e] != sq);
    rEnd[e] = flag[e] && (!nflag[e] || nseq[e] != sq);
    laneStarts += rStart[e] ? 1 : 0;
  }
  int ridx = mm_wave_excl_scan(laneStarts);                       // runs that started before this lane
  const int nRuns = mm_wave_sum(laneStarts);
  if (nRuns == 0) return 0;
  if (nRuns > MM_FUSE_MAXRUNS) return -1;
  int myRun[R];
#pragma unroll
  for (int e = 0; e < R; e++) {
    if (rStart[e]) { sc.run[ridx].seq = (int32_t)(gk[e] >> 32); sc.run[ridx].start = (int32_t)(uint32_t)gk[e]; sc.run[ridx].isize = 0; ridx++; }
    myRun[e] = ridx - 1;
  }
  __threadfence_block();
#pragma unroll
  for (int e = 0; e < R; e++) {
    if (flag[e]) atomicMax(&sc.run[myRun[e]].isize, sc.v[lane * R + e]);
    if (rEnd[e]) sc.run[myRun[e]].end = (int32_t)(uint32_t)gk[e];
  }
  __threadfence_block();
  // join runs closer than segLength (computeMap.hpp:1102-1115); uniform across the wave, compacted in place
  int nOut = 0;
  L1Run pend = sc.run[0];
  for (int r = 1; r < nRuns; r++) {
    const L1Run x = sc.run[r];
    if (x.seq != pend.seq || x.start > pend.end + segLength) {
      __threadfence_block();
      if (lane == 0) sc.run[nOut] = pend;
      nOut++; pend = x;
    } else { pend.end = x.end; pend.isize = x.isize > pend.isize ? x.isize : pend.isize; }
  }
  __threadfence_block();
  if (lane == 0) sc.run[nOut] = pend;
  nOut++;
  __threadfence_block();
  return nOut;
}

// ---------------------------------------------------------------------------------------------
// k_lookup_l1: one wave per fragment (4 per workgroup, no workgroup barrier).
//   * probes the s sketch hashes in the open-addressing table (key -> offset,count,frequent)
//   * drops frequent seeds and compacts the sketch (Q.minmerTableQuery / Q.sketchSize, computeMap.hpp:834-839)
//   * FAST (<= 128 interval points, no skip_prefix grouping): gathers the points into LDS, sorts them in registers
//     (ascending packed key == (seqId, pos, CLOSE-before-OPEN), computeMap.hpp:885-907) and runs L1 on the spot --
//     the points never touch HBM
//   * otherwise (many points, -Y groups, a position group spanning contigs, or keepPoints for the parity API): reserves
//     slots in the global point buffer, gathers there and queues the fragment for k_sort_points_* + k_l1_sweep.
// ---------------------------------------------------------------------------------------------
// The table values of the sketch entries [base, base + 256) of a fragment, four per lane (entry base + u * 64 + lane): found[u] / val[u].
// Four sub-rounds are in flight together: their hash loads, filter tests / tag loads and first table slots are independent, so one
// memory round trip serves all of them instead of four.  Shared by k_lookup_l1 and k_gather_points.
template <bool TAGS>
__device__ __forceinline__ void mm_probe4(const SeedTable& T, const uint64_t* __restrict__ skHash, size_t fo, int cnt, int base, int lane,
                                          uint64_t (&h)[4], bool (&act)[4], bool (&found)[4], uint64_t (&val)[4]) {
  const HtSlot* __restrict__ ht = T.ht; const uint64_t htMask = T.mask;
  bool open[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { const int r = base + u * 64 + lane; act[u] = r < cnt; h[u] = act[u] ? skHash[fo + r] : 0ull; }
  if constexpr (!TAGS) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      open[u] = act[u];
      if (T.filterMask && act[u]) { const uint64_t fb = mm_filter_bits(h[u]); open[u] = (T.filter[mm_filter_word(h[u], T.filterMask)] & fb) == fb; }
    }
    HtSlot sl[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { sl[u].key = MM_EMPTY; sl[u].val = 0; if (open[u]) sl[u] = ht[h[u] & htMask]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      found[u] = false; val[u] = 0;
      uint64_t slot = h[u] & htMask;
      while (open[u]) {                                            // first slot already loaded; further slots are rare (load <= 0.5)
        if (sl[u].key == h[u]) { found[u] = true; val[u] = sl[u].val; break; }
        if (sl[u].key == MM_EMPTY) break;
        slot = (slot + 1) & htMask;
        sl[u] = ht[slot];
      }
    }
  } else {
    // Tagged table (human-scale index).  One 16-byte load per seed fetches the tag bytes of its home bucket, out of an array 1/16 the
    // size of the table.  A slot is fetched only where a tag matches (a present seed, or one absent seed in ~40 by chance); an absent
    // seed ends at the first bucket with a free slot: its own, but for ~0.03 %.
    const uint8_t* __restrict__ tags = T.tags;
    uint4 tg[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { tg[u] = make_uint4(0u, 0u, 0u, 0u); if (act[u]) tg[u] = *(const uint4*)(tags + ((h[u] & htMask) & ~(uint64_t)(MM_TAG_BUCKET - 1))); }
    uint32_t cand[4]; bool emp[4]; HtSlot sl[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      mm_tag_scan(tg[u], mm_seed_tag(h[u]), cand[u], emp[u]);
      if (!act[u]) { cand[u] = 0; emp[u] = true; }
      sl[u].key = MM_EMPTY; sl[u].val = 0;
      if (cand[u]) sl[u] = ht[((h[u] & htMask) & ~(uint64_t)(MM_TAG_BUCKET - 1)) + (uint32_t)__builtin_ctz(cand[u])];   // the first candidates of all four sub-rounds travel together
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      found[u] = false; val[u] = 0;
      uint64_t b = (h[u] & htMask) & ~(uint64_t)(MM_TAG_BUCKET - 1);
      uint32_t cd = cand[u]; bool em = emp[u]; bool have = cd != 0;  // `have`: sl[u] holds the slot of cd's lowest bit
      while (cd || !em) {
        while (cd) {
          const uint32_t i = (uint32_t)__builtin_ctz(cd); cd &= cd - 1u;
          const HtSlot x = have ? sl[u] : ht[b + i];
          have = false;
          if (x.key == h[u]) { found[u] = true; val[u] = x.val; cd = 0; em = true; }
        }
        if (!em) {                                                   // a full bucket without the key: on to the next one
          b = (b + MM_TAG_BUCKET) & htMask;
          mm_tag_scan(*(const uint4*)(tags + b), mm_seed_tag(h[u]), cd, em);
        }
      }
    }
  }
}
// the long point runs of one probing sub-round of the fused kernel (lanes in mLong own a seed with more than 8 points), copied by the whole
// wave into the fragment's LDS list from their third point on; returns this lane's count of points that passed the seqId filters.  Not
// inlined: inside k_lookup_l1 the loop cost the uniform north_star workload 1.5 ms of registers and code layout (profiles/NOTES.md, round 5)
__device__ __noinline__ int mm_fused_long_runs(uint64_t mLong, int c, int my, uint64_t src, uint64_t* __restrict__ dst, const uint64_t* __restrict__ ptKeys,
                                               int self, int seqCounter, int skipSelf, int lowerTri, int lane) {
  int nValid = 0;
  while (mLong) {
    const int l = (int)__builtin_ctzll(mLong); mLong &= mLong - 1ull;
    const int cL = __shfl(c, l), myL = __shfl(my, l);
    const uint64_t srcL = ((uint64_t)(uint32_t)__shfl((int)(src >> 32), l) << 32) | (uint32_t)__shfl((int)(uint32_t)src, l);
    for (int j = 2 + lane; j < cL; j += 64) {
      uint64_t key = ptKeys[srcL + j];
      const int seqId = (int)(key >> 33);
      bool drop = false;
      if (skipSelf && seqId == self) drop = true;
      if (lowerTri && !(seqCounter > seqId)) drop = true;
      if (drop) key = MM_EMPTY; else nValid++;
      dst[myL + j] = key;
    }
  }
  return nValid;
}
#define MM_LOOKUP_WPB 4             // waves (= fragments) per workgroup
#define MM_MID_MAXSKETCH 512        // k_lookup_mid (below): sketch sizes it serves,
#define MM_MID_MAXSEEDS 256         // surviving seeds WITH points whose runs its LDS lists (a fragment with more goes to the HBM path),
#define MM_MID_MAXKEEP 1024         // interval points that may survive its filter (sorted in registers, 16 per lane),
#define MM_MID_MAXPTS 16384         // and interval points of a fragment it takes (8 192 intervals: its 16-bit bin counters cannot overflow)
#define MM_L1_REGIONS 64            // L1 output cursors: a same-address atomic costs ~10 ns, so fragments spread over 64 of them
#define MM_L1_CURSOR_STRIDE 32      // u64 words between cursors (256 bytes)
template <int MAXPTS, bool TAGS>
__global__ void __launch_bounds__(MM_LOOKUP_WPB * 64, MAXPTS <= 128 ? 8 : MAXPTS <= 256 ? 6 : 5)
k_lookup_l1(int nFrags, int s, const DFrag* __restrict__ frags,
            const uint64_t* __restrict__ skHash, const int8_t* __restrict__ skStrand, const uint32_t* __restrict__ skCount,
            const SeedTable T, const uint64_t* __restrict__ ptKeys,
            const int32_t* __restrict__ readSelf, int seqCounterBase, MapFlags fl, int keepPoints,
            uint64_t* __restrict__ qHash, int8_t* __restrict__ qStrand,
            mm_frag_stats* __restrict__ stats, int64_t* __restrict__ ptOff, unsigned long long ptsCap,
            const int32_t* __restrict__ minHitsTab, const int32_t* __restrict__ cutoffs, int nCutoffs, int segLength,
            mm_l1_candidate* __restrict__ l1, unsigned long long regionCap, unsigned long long* __restrict__ l1Cursors,
            int64_t* __restrict__ l1Off, int32_t* __restrict__ bigList, int32_t* __restrict__ midList,
            unsigned long long* __restrict__ counters /* [0] point cursor [1] pts overflow [3] l1 overflow [7] big count [16] mid count */) {
  typedef FuseScratchT<MAXPTS> FuseScratch;
  __shared__ FuseScratch scratch[MM_LOOKUP_WPB];
  const int wv = threadIdx.x >> 6;
  const int f = blockIdx.x * MM_LOOKUP_WPB + wv;
  const bool live = f < nFrags;
  FuseScratch& sc = scratch[wv];
  const int lane = (int)mm_lane();
  int nValid = 0, nOut = -1, outIdx = 0, P = 0, cnt = 0;
  uint64_t lastHash = 0;
  if (live) {
  cnt = (int)skCount[f];
  const size_t fo = (size_t)f * s;
  // per-read metadata early: nothing below depends on the probes to fetch it
  const int readId = frags[f].readId;
  const int self = readSelf[readId], seqCounter = seqCounterBase + readId;
  lastHash = cnt ? skHash[fo + cnt - 1] : 0ull;
  // The interval points are gathered into LDS batch by batch (256 sketch entries are probed at a time), as long as they fit (MAXPTS):
  // a sketch of more than 256 entries stays on the fused path too.  The order of the points is irrelevant: they are sorted next.
  bool fuseOk = !keepPoints && !fl.skipPrefix;                     // wave-uniform
  int done = 0;                                                    // points in sc.a so far
  auto put = [&](int at, uint64_t key) {
    const int seqId = (int)(key >> 33);
    bool drop = false;
    if (fl.skipSelf && seqId == self) drop = true;
    if (fl.lowerTri && !(seqCounter > seqId)) drop = true;
    if (drop) key = MM_EMPTY; else nValid++;
    sc.a[at] = key;
  };
  bool anyDrop = false;                                            // a frequent seed has been removed so far (wave-uniform)
  for (int base = 0; base < cnt; base += 256) {
    uint64_t h[4], val[4]; bool act[4], found[4];
    mm_probe4<TAGS>(T, skHash, fo, cnt, base, lane, h, act, found, val);
    // The sketch after frequent-seed removal goes to qHash/qStrand only if a seed was removed: otherwise it equals the raw sketch, and
    // readers (k_l2_locate, mm_query_sketch_download) take that instead (rawSketchSize == sketchSize in the fragment's stats).  In a
    // sketch of several batches the first removal back-fills the batches before it, which were the raw sketch unchanged.
    {
      bool drop = false;
#pragma unroll
      for (int u = 0; u < 4; u++) drop |= act[u] && found[u] && (val[u] & 1ull);
      if (__ballot(drop) != 0 && !anyDrop) {
        for (int i = lane; i < base; i += 64) { qHash[fo + i] = skHash[fo + i]; qStrand[fo + i] = skStrand[fo + i]; }
        anyDrop = true;
      }
    }
    int cU[4]; uint64_t src[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = base + u * 64 + lane;
      const bool keep = act[u] && !(found[u] && (val[u] & 1ull));
      const uint64_t m = __ballot(keep);
      if (keep && anyDrop) {
        const int idx = outIdx + (int)mm_popc_below(m);
        qHash[fo + idx] = h[u]; qStrand[fo + idx] = skStrand[fo + r];
      }
      const bool kf = keep && found[u];
      src[u] = val[u] >> 24;
      cU[u] = kf ? (int)((val[u] >> 1) & 0x7fffffull) : 0;
      outIdx += __popcll(m);
    }
    const int Pb = mm_wave_sum(cU[0] + cU[1] + cU[2] + cU[3]);
    P += Pb;
    if (fuseOk && Pb > 0) {
      if (done + Pb > MAXPTS) fuseOk = false;
      else {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int c = cU[u];
          const int my = done + mm_wave_excl_scan(c);
          uint64_t k0 = MM_EMPTY, k1 = MM_EMPTY;
          if (c > 0) k0 = ptKeys[src[u]];
          if (c > 1) k1 = ptKeys[src[u] + 1];
          if (c > 0) put(my, k0);
          if (c > 1) put(my + 1, k1);
          // the rest of a short run by the lane that owns the seed, of a long one (a seed of a repeat family: up to hundreds of points
          // that still fit the fused list) by the whole wave, as in mm_gather_points
          const bool longRun = c > 8;
          if (!longRun) for (int j = 2; j < c; j++) put(my + j, ptKeys[src[u] + j]);
          const uint64_t mLong = mm_ballot(longRun);
          if (mLong) nValid += mm_fused_long_runs(mLong, c, my, src[u], sc.a, ptKeys, self, seqCounter, fl.skipSelf, fl.lowerTri, lane);   // (out of line: the hot path keeps its registers)
          done += mm_wave_sum(c);
        }
      }
    }
  }
  const int minHits0 = outIdx > 0 ? minHitsTab[outIdx] : 0;
  if (fuseOk && minHits0 > 0) {
    if (P == 0) nOut = 0;
    else {
      nValid = mm_wave_sum(nValid);
      const int padTo = P <= 64 ? 64 : P <= 128 ? 128 : P <= 256 ? 256 : 512;
      for (int j = P + lane; j < padTo; j += 64) sc.a[j] = MM_EMPTY;
      __threadfence_block();
      if (P <= 64) {
        uint64_t k[1] = {sc.a[lane]};
        mm_wave_bitonic<1>(k, lane);
        nOut = mm_l1_fused<1>(k, sc, outIdx, minHits0, fl.hg, cutoffs, nCutoffs, s, segLength, lane);
      } else if (P <= 128) {
        uint64_t k[2] = {sc.a[lane * 2], sc.a[lane * 2 + 1]};
        mm_wave_bitonic<2>(k, lane);
        nOut = mm_l1_fused<2>(k, sc, outIdx, minHits0, fl.hg, cutoffs, nCutoffs, s, segLength, lane);
      } else if constexpr (MAXPTS >= 256) {
        if (P <= 256) {
          uint64_t k[4] = {sc.a[lane * 4], sc.a[lane * 4 + 1], sc.a[lane * 4 + 2], sc.a[lane * 4 + 3]};
          mm_wave_bitonic<4>(k, lane);
          nOut = mm_l1_fused<4>(k, sc, outIdx, minHits0, fl.hg, cutoffs, nCutoffs, s, segLength, lane);
        } else if constexpr (MAXPTS >= 512) {
          uint64_t k[8];
#pragma unroll
          for (int e = 0; e < 8; e++) k[e] = sc.a[lane * 8 + e];
          mm_wave_bitonic<8>(k, lane);
          nOut = mm_l1_fused<8>(k, sc, outIdx, minHits0, fl.hg, cutoffs, nCutoffs, s, segLength, lane);
        }
      }
    }
  }
  // more points than the registers sort, nothing else in the way: k_lookup_mid filters them in LDS first (the fragment stays off the HBM
  // path unless more than 512 of its points can matter to L1)
  const bool toMid = nOut < 0 && midList && !fuseOk && !keepPoints && !fl.skipPrefix && minHits0 > 1 && cnt <= MM_MID_MAXSKETCH && P <= MM_MID_MAXPTS;
  if (toMid) {
    if (lane == 0) {
      mm_frag_stats st;
      st.rawSketchSize = cnt; st.sketchSize = outIdx; st.maxHash = lastHash; st.nPoints = 0; st.nL1 = 0; stats[f] = st;
      ptOff[2 * f] = 0; ptOff[2 * f + 1] = 0; l1Off[f] = 0;
      midList[atomicAdd(&counters[16], 1ull)] = f;
    }
  } else if (nOut < 0) {
    // slow path: the fragment's points go to HBM -- gathered by k_gather_points (which looks the seeds up once more: this kernel keeps no
    // per-seed value beyond the batch it is probing), sorted and swept by the follow-up kernels.  Here: the slots (a power of two above
    // 64 for the sorters) and the queue entry.
    int slots = P;
    if (P > 64) { slots = 128; while (slots < P) slots <<= 1; }
    else if (keepPoints == 2 && P > 0) { slots = 2; while (slots < P) slots <<= 1; }      // windowed mode: every list goes through the LDS / HBM sorters
    if (lane == 0) {
      unsigned long long off = 0;
      if (slots > 0) off = atomicAdd(&counters[0], (unsigned long long)slots);
      bool ok = true;
      if (slots > 0 && off + (unsigned long long)slots > ptsCap) { ok = false; atomicOr(&counters[1], 1ull); }
      mm_frag_stats st;
      st.rawSketchSize = cnt; st.sketchSize = outIdx; st.maxHash = lastHash; st.nPoints = 0; st.nL1 = 0; stats[f] = st;
      ptOff[2 * f] = (int64_t)off; ptOff[2 * f + 1] = ok ? (int64_t)slots : 0; l1Off[f] = 0;
      if (slots > 0 && ok) bigList[atomicAdd(&counters[7], 1ull)] = f;
    }
  }
  }  // live
  if (live && nOut >= 0) {                                         // fast path complete: emit the candidates
    // region f mod 64 of the L1 buffer, filled from its own cursor; k_l1_compact closes the gaps afterwards
    const int region = f & (MM_L1_REGIONS - 1);
    unsigned long long at = 0;
    if (nOut > 0) {
      if (lane == 0) at = atomicAdd(&l1Cursors[(size_t)region * MM_L1_CURSOR_STRIDE], (unsigned long long)nOut);
      at = ((unsigned long long)(uint32_t)__shfl((int)(at >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)at, 0);
      if (at + (unsigned long long)nOut > regionCap) { if (lane == 0) atomicOr(&counters[3], 1ull); nOut = 0; }
    }
    const unsigned long long base = (unsigned long long)region * regionCap + at;
    for (int i = lane; i < nOut; i += 64) {
      const L1Run x = sc.run[i];
      mm_l1_candidate o; o.frag = f; o.seqId = x.seq; o.rangeStartPos = x.start; o.rangeEndPos = x.end; o.intersectionSize = x.isize;
      l1[base + i] = o;
    }
    if (lane == 0) {
      mm_frag_stats st;
      st.rawSketchSize = cnt; st.sketchSize = outIdx; st.maxHash = lastHash; st.nPoints = nValid; st.nL1 = nOut; stats[f] = st;
      ptOff[2 * f] = 0; ptOff[2 * f + 1] = 0; l1Off[f] = (int64_t)base;
    }
  }
}

// k_gather_points: the interval points of the fragments k_lookup_l1 queued (more points than its registers hold, -Y reference groups,
// --noSplit, MM_OPT_KEEP_POINTS) -> the slots it reserved for them in HBM, one wave per queued fragment: the sketch is looked up again,
// batch by batch, and every surviving seed's point run is copied (getSeedIntervalPoints, computeMap.hpp:857-912, with the seqId
// filters of :891-896).  ids (windowed mode): the seed every point came from.
template <bool TAGS>
__global__ void __launch_bounds__(256)
k_gather_points(int nList, const int32_t* __restrict__ list, int s, const DFrag* __restrict__ frags, const uint64_t* __restrict__ skHash, const uint32_t* __restrict__ skCount,
                const SeedTable T, const uint64_t* __restrict__ ptKeys, const int32_t* __restrict__ refGroup, const int32_t* __restrict__ readGroup, const int32_t* __restrict__ readSelf,
                int seqCounterBase, MapFlags fl, mm_frag_stats* __restrict__ stats, const int64_t* __restrict__ ptOff, uint64_t* __restrict__ pts, uint16_t* __restrict__ ptIds,
                int32_t* __restrict__ ptKept, const unsigned long long* __restrict__ nDev) {
  if (nDev) nList = (int)*nDev;
  for (int li = blockIdx.x * 4 + (threadIdx.x >> 6); li < nList; li += gridDim.x * 4) {
  const int f = list[li];
  const int lane = (int)mm_lane();
  const int64_t off = ptOff[2 * f]; const int slots = (int)ptOff[2 * f + 1];
  if (slots <= 0) continue;
  const int cnt = (int)skCount[f];
  const size_t fo = (size_t)f * s;
  const int readId = frags[f].readId;
  const int rg = readGroup[readId], self = readSelf[readId], seqCounter = seqCounterBase + readId;
  int at = 0, nValid = 0;
  uint16_t* idDst = ptIds ? ptIds + off : nullptr;
  for (int base = 0; base < cnt; base += 256) {
    uint64_t h[4], val[4]; bool act[4], found[4];
    mm_probe4<TAGS>(T, skHash, fo, cnt, base, lane, h, act, found, val);
#pragma unroll
    for (int u = 0; u < 4; u++) val[u] = (act[u] && found[u] && !(val[u] & 1ull)) ? val[u] : 0ull;
    nValid += mm_gather_points(pts + off, at, base >> 6, 4, [&](int rd) { return rd == 0 ? val[0] : rd == 1 ? val[1] : rd == 2 ? val[2] : val[3]; },
                               ptKeys, refGroup, rg, self, seqCounter, fl, lane, idDst);
  }
  for (int j = at + lane; j < slots; j += 64) pts[off + j] = MM_EMPTY;
  if (lane == 0) { stats[f].nPoints = nValid; ptKept[f] = nValid; }      // ptKept: the points the L1 kernels will find at the head of the sorted list
  }
}

// k_filter_points: drops, BEFORE the sort, the interval points of a queued fragment that cannot matter to computeL1CandidateRegions
// (computeMap.hpp:916-1116, windowLen == 0).  Against a repeat-rich reference a fragment brings hundreds of scattered single hits with it
// (the lists that overflow the fused kernel are mostly those): the bitonic sorters paid for them with two thirds of a pass
// (profiles/r13a: 75 of 183 ms at the repeat-rich north_star workload).  Rule (tests/l1filter.py is its model, tests/test_l1_point_filter.py
// checks it against the oracle's literal L1 on fuzzed point sets): the gathered list is a sequence of (OPEN, CLOSE) pairs, one per
// interval [o, c) of a contig; the overlap count of a position is at most the number of intervals that intersect its 4 096-position bin, so
// an interval none of whose bins is intersected by minimumHits intervals covers no position that reaches minimumHits -- it neither starts,
// ends nor raises a candidate -- and goes, unless it holds the first or the last point of its contig: the reference's sweep groups points
// by `pos` alone (:967, :1047-1051), so the last point of one contig and the first of the next can share a group, and with both boundary
// groups kept whole that seam behaves as it did.  Bins are counted in a hashed LDS table (collisions only raise counts: more is kept, never
// less); a fragment with an interval over more than MM_FILT_MAXSPAN bins or with more contigs than the per-wave table holds is left as
// it is.  One wave per queued fragment, in place; the list's length (ptOff[2f+1], what the sorters go by) shrinks with it.
#define MM_FILT_SHIFT 12
#define MM_FILT_SLOTS 2048
#define MM_FILT_CONTIGS 128
#define MM_FILT_MAXSPAN 64
__device__ __forceinline__ uint32_t mm_bin_slot(uint32_t seq, uint32_t b) { return ((seq * 0x9E3779B1u + b * 0x85EBCA77u) >> 7) & (MM_FILT_SLOTS - 1); }
__global__ void __launch_bounds__(256)
k_filter_points(int nList, const int32_t* __restrict__ list, const mm_frag_stats* __restrict__ stats, const int32_t* __restrict__ minHitsTab,
                int64_t* __restrict__ ptOff, uint64_t* __restrict__ pts, int32_t* __restrict__ ptKept, const unsigned long long* __restrict__ nDev) {
  __shared__ uint32_t binCntAll[4][MM_FILT_SLOTS];
  __shared__ int32_t cKeyAll[4][MM_FILT_CONTIGS];
  __shared__ uint32_t cMinAll[4][MM_FILT_CONTIGS], cMaxAll[4][MM_FILT_CONTIGS];
  const int wv = threadIdx.x >> 6, lane = (int)mm_lane();
  uint32_t* binCnt = binCntAll[wv]; int32_t* cKey = cKeyAll[wv]; uint32_t* cMin = cMinAll[wv]; uint32_t* cMax = cMaxAll[wv];
  if (nDev) nList = (int)*nDev;
  for (int li = blockIdx.x * 4 + wv; li < nList; li += gridDim.x * 4) {
    const int f = list[li];
    const int64_t off = ptOff[2 * f]; const int slots = (int)ptOff[2 * f + 1];
    if (slots <= 2) continue;
    const mm_frag_stats st = stats[f];
    const int minHits = st.sketchSize > 0 ? minHitsTab[st.sketchSize] : 0;
    if (minHits <= 1) continue;                                    // every interval reaches a count of 1 (minimumHits 0: the literal sweep takes the list whole)
    const int pairs = slots >> 1;
    for (int i = lane; i < MM_FILT_SLOTS; i += 64) binCnt[i] = 0u;
    for (int i = lane; i < MM_FILT_CONTIGS; i += 64) { cKey[i] = -1; cMin[i] = 0xFFFFFFFFu; cMax[i] = 0u; }
    __threadfence_block();
    uint64_t* a = pts + off;
    bool giveUp = false;
    for (int i = lane; i < pairs; i += 64) {
      const uint64_t O = a[2 * i], C = a[2 * i + 1];
      if (O == MM_EMPTY) continue;
      const uint32_t seq = (uint32_t)(O >> 33), o = (uint32_t)(O >> 1), c = (uint32_t)(C >> 1);
      const uint32_t b0 = o >> MM_FILT_SHIFT, b1 = (c - 1u) >> MM_FILT_SHIFT;
      if (C == MM_EMPTY || c <= o || b1 - b0 >= (uint32_t)MM_FILT_MAXSPAN) { giveUp = true; continue; }
      for (uint32_t b = b0; b <= b1; b++) atomicAdd(&binCnt[mm_bin_slot(seq, b)], 1u);
      uint32_t sl = (seq * 0x9E3779B1u) >> 25;                       // 7 bits
      int tries = 0;
      for (; tries < MM_FILT_CONTIGS; tries++) {
        const int32_t prev = atomicCAS(&cKey[sl], -1, (int32_t)seq);
        if (prev == -1 || prev == (int32_t)seq) { atomicMin(&cMin[sl], o); atomicMax(&cMax[sl], c); break; }
        sl = (sl + 1u) & (MM_FILT_CONTIGS - 1);
      }
      if (tries == MM_FILT_CONTIGS) giveUp = true;
    }
    __threadfence_block();
    if (mm_ballot(giveUp)) continue;                               // left as it is (ptKept[f] is what k_gather_points wrote)
    int cursor = 0;                                                // pairs kept so far (wave-uniform)
    for (int base = 0; base < pairs; base += 64) {
      const int i = base + lane;
      uint64_t O = MM_EMPTY, C = MM_EMPTY;
      if (i < pairs) { O = a[2 * i]; C = a[2 * i + 1]; }
      bool keep = false;
      if (O != MM_EMPTY) {
        const uint32_t seq = (uint32_t)(O >> 33), o = (uint32_t)(O >> 1), c = (uint32_t)(C >> 1);
        const uint32_t b0 = o >> MM_FILT_SHIFT, b1 = (c - 1u) >> MM_FILT_SHIFT;
        for (uint32_t b = b0; b <= b1; b++) keep = keep || binCnt[mm_bin_slot(seq, b)] >= (uint32_t)minHits;
        uint32_t sl = (seq * 0x9E3779B1u) >> 25;
        while (cKey[sl] != (int32_t)seq) sl = (sl + 1u) & (MM_FILT_CONTIGS - 1);
        keep = keep || o == cMin[sl] || c == cMax[sl];               // the contig's first / last position group stays whole
      }
      const uint64_t m = mm_ballot(keep);
      // in place: every lane of this round has read its pair before anybody writes, and a pair only ever moves down
      if (keep) { const int at = cursor + (int)mm_popc_below(m); a[2 * at] = O; a[2 * at + 1] = C; }
      cursor += (int)__popcll(m);
    }
    const int kept = 2 * cursor;
    int newSlots = kept;
    if (kept > 64) { newSlots = 128; while (newSlots < kept) newSlots <<= 1; }
    for (int j = kept + lane; j < newSlots; j += 64) a[j] = MM_EMPTY;
    if (lane == 0) { ptOff[2 * f + 1] = (int64_t)newSlots; ptKept[f] = kept; }
    __threadfence_block();
  }
}

// k_lookup_mid: the fragments whose interval points outnumber what k_lookup_l1 sorts in registers (against a repeat-rich reference 31 % of
// the north_star fragments: hundreds of scattered single hits each) -- WITHOUT the trip through HBM (k_gather_points -> k_filter_points ->
// sorters -> k_l1_stream: 33 of that workload's 125 ms).  One wave per queued fragment: the seeds are probed once more and their table
// answers kept in LDS; a first sweep over the seeds' point runs counts, per 4 096-position bin, the intervals that intersect it (and per
// contig the first and last position), a second sweep keeps the intervals k_filter_points' rule keeps (same bins, same hashed table, same
// boundary rule: tests/l1filter.py is the model) and writes them into the LDS list; at most 512 survivors are sorted in registers and
// swept by mm_l1_fused exactly as in k_lookup_l1, and the candidates go to the same region cursors.  A fragment with more survivors, with
// an interval the filter gives up on, or whose sorted list needs the literal sweep is queued for the HBM path as k_lookup_l1 would have.
// 513 .. 1 024 surviving points: sixteen per lane.  Out of line: the common sizes keep their registers.
__device__ __noinline__ int mm_mid_sort16(FuseScratchT<MM_MID_MAXKEEP>& sc, int sketchSizeQ, int minHits, int hg, const int32_t* __restrict__ cutoffs, int nCutoffs,
                                          int sParam, int segLength, int lane) {
  uint64_t k[16];
#pragma unroll
  for (int e = 0; e < 16; e++) k[e] = sc.a[lane * 16 + e];
  mm_wave_bitonic<16>(k, lane);
  return mm_l1_fused<16>(k, sc, sketchSizeQ, minHits, hg, cutoffs, nCutoffs, sParam, segLength, lane);
}

template <bool TAGS>
__global__ void __launch_bounds__(64)
k_lookup_mid(int nList, const unsigned long long* __restrict__ nDev, const int32_t* __restrict__ list, int s, const DFrag* __restrict__ frags,
             const uint64_t* __restrict__ skHash, const uint32_t* __restrict__ skCount, const SeedTable T, const uint64_t* __restrict__ ptKeys,
             const int32_t* __restrict__ readSelf, int seqCounterBase, MapFlags fl, mm_frag_stats* __restrict__ stats, int64_t* __restrict__ ptOff,
             unsigned long long ptsCap, const int32_t* __restrict__ minHitsTab, const int32_t* __restrict__ cutoffs, int nCutoffs, int segLength,
             mm_l1_candidate* __restrict__ l1, unsigned long long regionCap, unsigned long long* __restrict__ l1Cursors, int64_t* __restrict__ l1Off,
             int32_t* __restrict__ bigList, unsigned long long* __restrict__ counters) {
  // LDS of the one wave: the survivors' list (FuseScratch::a) lives through both phases; the seeds' runs, the bins and the contig extents
  // are dead when the sort starts and share their bytes with what mm_l1_fused needs then (FuseScratch::v, ::run)
  typedef FuseScratchT<MM_MID_MAXKEEP> FuseScratch;
  struct Phase1 {
    uint64_t src[MM_MID_MAXSEEDS];                                 // first point of every surviving seed's run
    uint32_t pre[MM_MID_MAXSEEDS + 1];                             // intervals (point pairs) of the seeds before it
    uint32_t binCnt[MM_FILT_SLOTS / 2];                            // two 16-bit counters per word
    int32_t cKey[MM_FILT_CONTIGS]; uint32_t cMin[MM_FILT_CONTIGS], cMax[MM_FILT_CONTIGS];
  };
  constexpr size_t kTail = sizeof(FuseScratch) - offsetof(FuseScratch, v);
  constexpr size_t kBytes = offsetof(FuseScratch, v) + (sizeof(Phase1) > kTail ? sizeof(Phase1) : kTail);
  __shared__ __attribute__((aligned(16))) unsigned char smem[kBytes];
  FuseScratch& sc = *(FuseScratch*)smem;
  Phase1& ph = *(Phase1*)(smem + offsetof(FuseScratch, v));
  const int lane = (int)mm_lane();
  if (nDev) nList = (int)*nDev;
  for (int li = blockIdx.x; li < nList; li += gridDim.x) {
    const int f = list[li];
    const int cnt = (int)skCount[f];
    const size_t fo = (size_t)f * s;
    const int readId = frags[f].readId;
    const int self = readSelf[readId], seqCounter = seqCounterBase + readId;
    const mm_frag_stats st0 = stats[f];
    const int outIdx = st0.sketchSize;
    const int minHits = outIdx > 0 ? minHitsTab[outIdx] : 0;
    for (int i = lane; i < MM_FILT_SLOTS / 2; i += 64) ph.binCnt[i] = 0u;
    for (int i = lane; i < MM_FILT_CONTIGS; i += 64) { ph.cKey[i] = -1; ph.cMin[i] = 0xFFFFFFFFu; ph.cMax[i] = 0u; }
    // the seeds once more; those with points are listed densely: run start + intervals before it
    int nSeeds = 0, nPairs = 0, P = 0; bool giveUp = false;
    for (int base = 0; base < cnt; base += 256) {
      uint64_t h[4], val[4]; bool act[4], found[4];
      mm_probe4<TAGS>(T, skHash, fo, cnt, base, lane, h, act, found, val);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool kf = act[u] && found[u] && !(val[u] & 1ull);
        const int c = kf ? (int)((val[u] >> 1) & 0x7fffffull) : 0;
        if (c & 1) giveUp = true;                                  // (never: a hash's points are its intervals' two ends)
        const uint64_t m = mm_ballot(c > 0);
        const int at = nSeeds + (int)mm_popc_below(m);
        const int before = nPairs + mm_wave_excl_scan(c >> 1);
        if (c > 0 && at < MM_MID_MAXSEEDS) { ph.src[at] = val[u] >> 24; ph.pre[at] = (uint32_t)before; }
        nSeeds += (int)__popcll(m); nPairs += mm_wave_sum(c >> 1); P += mm_wave_sum(c);
      }
    }
    if (nSeeds > MM_MID_MAXSEEDS) { giveUp = true; nSeeds = MM_MID_MAXSEEDS; nPairs = 0; }
    if (lane == 0) ph.pre[nSeeds] = (uint32_t)nPairs;
    __threadfence_block();
    // interval g of the fragment = pair (g - pre[j]) of seed j's run, j by binary search: every lane of every round has one
    auto pairAt = [&](int g, uint64_t& O, uint64_t& C) {
      int lo = 0, hi = nSeeds - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)ph.pre[mid] <= g) lo = mid; else hi = mid - 1; }
      const ulonglong2 oc = *(const ulonglong2*)(ptKeys + ph.src[lo] + 2 * (uint64_t)(g - (int)ph.pre[lo]));
      O = oc.x; C = oc.y;
    };
    auto dropped = [&](uint64_t key) {
      const int seqId = (int)(key >> 33);
      return (fl.skipSelf && seqId == self) || (fl.lowerTri && !(seqCounter > seqId));
    };
    // sweep 1: intervals per 4 096-position bin, first / last position per contig
    int nValid = 0;
    for (int g0 = 0; g0 < nPairs; g0 += 128) {
      uint64_t O[2], C[2]; bool on[2];
#pragma unroll
      for (int u = 0; u < 2; u++) { const int g = g0 + u * 64 + lane; on[u] = g < nPairs; O[u] = C[u] = MM_EMPTY; if (on[u]) pairAt(g, O[u], C[u]); }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (!on[u] || dropped(O[u])) continue;
        nValid += 2;
        const uint32_t seq = (uint32_t)(O[u] >> 33), o = (uint32_t)(O[u] >> 1), c = (uint32_t)(C[u] >> 1);
        const uint32_t b0 = o >> MM_FILT_SHIFT, b1 = (c - 1u) >> MM_FILT_SHIFT;
        if (c <= o || b1 - b0 >= (uint32_t)MM_FILT_MAXSPAN || (uint32_t)(C[u] >> 33) != seq) { giveUp = true; continue; }
        for (uint32_t b = b0; b <= b1; b++) { const uint32_t sl = mm_bin_slot(seq, b); atomicAdd(&ph.binCnt[sl >> 1], 1u << (16 * (sl & 1u))); }
        uint32_t sl = (seq * 0x9E3779B1u) >> 25;
        int tries = 0;
        for (; tries < MM_FILT_CONTIGS; tries++) {
          const int32_t prev = atomicCAS(&ph.cKey[sl], -1, (int32_t)seq);
          if (prev == -1 || prev == (int32_t)seq) { atomicMin(&ph.cMin[sl], o); atomicMax(&ph.cMax[sl], c); break; }
          sl = (sl + 1u) & (MM_FILT_CONTIGS - 1);
        }
        if (tries == MM_FILT_CONTIGS) giveUp = true;
      }
    }
    __threadfence_block();
    nValid = mm_wave_sum(nValid);
    bool fallback = mm_ballot(giveUp) != 0ull;
    // sweep 2: the intervals that can matter (k_filter_points' rule), into the LDS list
    int cursor = 0;                                                // pairs kept so far (wave-uniform)
    for (int g0 = 0; g0 < nPairs && !fallback; g0 += 128) {
      uint64_t O[2], C[2]; bool keep[2];
#pragma unroll
      for (int u = 0; u < 2; u++) { const int g = g0 + u * 64 + lane; keep[u] = g < nPairs; O[u] = C[u] = MM_EMPTY; if (keep[u]) pairAt(g, O[u], C[u]); }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (keep[u]) keep[u] = !dropped(O[u]);
        if (keep[u]) {
          const uint32_t seq = (uint32_t)(O[u] >> 33), o = (uint32_t)(O[u] >> 1), c = (uint32_t)(C[u] >> 1);
          const uint32_t b0 = o >> MM_FILT_SHIFT, b1 = (c - 1u) >> MM_FILT_SHIFT;
          bool k = false;
          for (uint32_t b = b0; b <= b1; b++) { const uint32_t sl = mm_bin_slot(seq, b); k = k || ((ph.binCnt[sl >> 1] >> (16 * (sl & 1u))) & 0xFFFFu) >= (uint32_t)minHits; }
          uint32_t sl = (seq * 0x9E3779B1u) >> 25;
          while (ph.cKey[sl] != (int32_t)seq) sl = (sl + 1u) & (MM_FILT_CONTIGS - 1);
          keep[u] = k || o == ph.cMin[sl] || c == ph.cMax[sl];
        }
        const uint64_t m = mm_ballot(keep[u]);
        const int k = (int)__popcll(m);
        if (cursor + k > MM_MID_MAXKEEP / 2) { fallback = true; break; }
        if (keep[u]) { const int at = cursor + (int)mm_popc_below(m); sc.a[2 * at] = O[u]; sc.a[2 * at + 1] = C[u]; }
        cursor += k;
      }
    }
    __threadfence_block();                                         // phase 1's LDS is free from here on
    int nOut = -1;
    if (!fallback) {
      const int K = 2 * cursor;
      if (K == 0) nOut = 0;
      else {
        const int padTo = K <= 64 ? 64 : K <= 128 ? 128 : K <= 256 ? 256 : K <= 512 ? 512 : 1024;
        for (int j = K + lane; j < padTo; j += 64) sc.a[j] = MM_EMPTY;
        __threadfence_block();
        if (K > 512) nOut = mm_mid_sort16(sc, outIdx, minHits, fl.hg, cutoffs, nCutoffs, s, segLength, lane);
        else if (K <= 64) { uint64_t k[1] = {sc.a[lane]}; mm_wave_bitonic<1>(k, lane); nOut = mm_l1_fused<1>(k, sc, outIdx, minHits, fl.hg, cutoffs, nCutoffs, s, segLength, lane); }
        else if (K <= 128) { uint64_t k[2] = {sc.a[lane * 2], sc.a[lane * 2 + 1]}; mm_wave_bitonic<2>(k, lane); nOut = mm_l1_fused<2>(k, sc, outIdx, minHits, fl.hg, cutoffs, nCutoffs, s, segLength, lane); }
        else if (K <= 256) {
          uint64_t k[4] = {sc.a[lane * 4], sc.a[lane * 4 + 1], sc.a[lane * 4 + 2], sc.a[lane * 4 + 3]};
          mm_wave_bitonic<4>(k, lane); nOut = mm_l1_fused<4>(k, sc, outIdx, minHits, fl.hg, cutoffs, nCutoffs, s, segLength, lane);
        } else {
          uint64_t k[8];
#pragma unroll
          for (int e = 0; e < 8; e++) k[e] = sc.a[lane * 8 + e];
          mm_wave_bitonic<8>(k, lane); nOut = mm_l1_fused<8>(k, sc, outIdx, minHits, fl.hg, cutoffs, nCutoffs, s, segLength, lane);
        }
      }
    }
    if (nOut < 0) {                                                // to the HBM path after all, as k_lookup_l1 queues a fragment
      int slots = 128; while (slots < P) slots <<= 1;
      if (lane == 0) {
        const unsigned long long off = atomicAdd(&counters[0], (unsigned long long)slots);
        bool ok = true;
        if (off + (unsigned long long)slots > ptsCap) { ok = false; atomicOr(&counters[1], 1ull); }
        ptOff[2 * f] = (int64_t)off; ptOff[2 * f + 1] = ok ? (int64_t)slots : 0; l1Off[f] = 0;
        if (ok) bigList[atomicAdd(&counters[7], 1ull)] = f;
      }
    } else {
      const int region = f & (MM_L1_REGIONS - 1);
      unsigned long long at = 0;
      if (nOut > 0) {
        if (lane == 0) at = atomicAdd(&l1Cursors[(size_t)region * MM_L1_CURSOR_STRIDE], (unsigned long long)nOut);
        at = ((unsigned long long)(uint32_t)__shfl((int)(at >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)at, 0);
        if (at + (unsigned long long)nOut > regionCap) { if (lane == 0) atomicOr(&counters[3], 1ull); nOut = 0; }
      }
      const unsigned long long base = (unsigned long long)region * regionCap + at;
      for (int i = lane; i < nOut; i += 64) {
        const L1Run x = sc.run[i];
        mm_l1_candidate o; o.frag = f; o.seqId = x.seq; o.rangeStartPos = x.start; o.rangeEndPos = x.end; o.intersectionSize = x.isize;
        l1[base + i] = o;
      }
      if (lane == 0) {
        mm_frag_stats st = st0;
        st.nPoints = nValid; st.nL1 = nOut; stats[f] = st;
        ptOff[2 * f] = 0; ptOff[2 * f + 1] = 0; l1Off[f] = (int64_t)base;
      }
    }
    __threadfence_block();
  }
}

// closes the gaps between the 64 regions of the L1 buffer: region r moves to its prefix position (into a second buffer), and
// the fragments' first-candidate offsets follow
struct L1Regions { unsigned long long prefix[MM_L1_REGIONS]; unsigned long long count[MM_L1_REGIONS]; };
// the regions' fill counts -> their prefix positions in the dense buffer (device-resident: no host round trip between the lookup kernel
// and the compaction) and the number of fused candidates (counters[2])
__global__ void __launch_bounds__(64)
k_l1_regions(const unsigned long long* __restrict__ l1Cursors, unsigned long long regionCap, L1Regions* __restrict__ R, unsigned long long* __restrict__ counters) {
  const int r = threadIdx.x;
  unsigned long long n = l1Cursors[(size_t)r * MM_L1_CURSOR_STRIDE];
  if (n > regionCap) n = regionCap;                              // an overflowed region (counters[3] is set): what was written
  const int ex = mm_wave_excl_scan((int)n);
  R->count[r] = n; R->prefix[r] = (unsigned long long)ex;
  if (r == 63) counters[2] = (unsigned long long)ex + n;
}
__global__ void __launch_bounds__(256)
k_l1_compact(const mm_l1_candidate* __restrict__ src, mm_l1_candidate* __restrict__ dst, unsigned long long regionCap, const L1Regions* __restrict__ R) {
  const int r = blockIdx.y;
  const unsigned long long n = R->count[r], at = R->prefix[r];
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256)
    dst[at + i] = src[(unsigned long long)r * regionCap + i];
}
__global__ void __launch_bounds__(256)
k_l1_fix_offsets(int nFrags, const mm_frag_stats* __restrict__ stats, int64_t* __restrict__ l1Off, unsigned long long regionCap, const L1Regions* __restrict__ R) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= nFrags || stats[f].nL1 <= 0) return;
  const int r = f & (MM_L1_REGIONS - 1);
  l1Off[f] = l1Off[f] - (int64_t)((unsigned long long)r * regionCap) + (int64_t)R->prefix[r];
}

// Steady-state passes read the L1 stage's overflow flags (counters[1] interval points, counters[3] L1 candidates) only when the whole
// pass is over, so the L2 stage must be kept off what an overflowed L1 stage left behind: a region that overflowed has never-written
// slots below its clamped count, the sweep path may have counted past the dense buffer, and more candidates than the previous pass
// sized the candidate-indexed buffers for (candCap) do not fit them either.  One thread, behind the last L1 kernel: any of these zeroes
// the candidate count -- every kernel of the L2 stage covers [0, counters[2]) and k_l2_select returns on a set flag -- and leaves (or
// sets) the flag that makes the host redo the pass the sized way.
__global__ void k_l1_gate(unsigned long long* __restrict__ counters, unsigned long long candCap) {
  if (counters[1] | counters[3]) counters[2] = 0ull;
  else if (counters[2] > candCap) { counters[3] = 1ull; counters[2] = 0ull; }
}

// ---------------------------------------------------------------------------------------------
// point sorters for the queued fragments (ascending packed key == ascending (seqId, pos, CLOSE-before-OPEN))
// ---------------------------------------------------------------------------------------------
// <= 512 points: one wave per fragment, bitonic network over the lanes' registers
#define MM_SORT_WAVECAP 512
// The kernels over the queued fragments take the list's length from the device when nDev is given (the launcher then has not read it
// back: steady-state passes, one host synchronisation per pass) and walk the list with whatever grid they were given.
__global__ void __launch_bounds__(256)
k_sort_points_wave(int nList, const unsigned long long* __restrict__ nDev, const int32_t* __restrict__ list, const int64_t* __restrict__ ptOff, uint64_t* __restrict__ pts) {
  if (nDev) nList = (int)*nDev;
  for (int li = blockIdx.x * 4 + (threadIdx.x >> 6); li < nList; li += gridDim.x * 4) {
    const int f = list[li];
    const int64_t off = ptOff[2 * f]; const int n = (int)ptOff[2 * f + 1];
    if (n <= 1 || n > MM_SORT_WAVECAP) continue;
    const int lane = (int)mm_lane();
    // up to 512 points in the registers of one wave (the network of the fused kernel): no LDS, no barrier.  What k_filter_points leaves of
    // a repeat-rich fragment's list is a few hundred points; the workgroup-per-fragment LDS sorter below took 16 ms per 0.6 M such lists
    if (n <= 64) {
      uint64_t k[1] = {lane < n ? pts[off + lane] : MM_EMPTY};
      mm_wave_bitonic<1>(k, lane);
      if (lane < n) pts[off + lane] = k[0];
    } else if (n <= 128) {
      uint64_t k[2];
#pragma unroll
      for (int r = 0; r < 2; r++) k[r] = lane * 2 + r < n ? pts[off + lane * 2 + r] : MM_EMPTY;
      mm_wave_bitonic<2>(k, lane);
#pragma unroll
      for (int r = 0; r < 2; r++) if (lane * 2 + r < n) pts[off + lane * 2 + r] = k[r];
    } else if (n <= 256) {
      uint64_t k[4];
#pragma unroll
      for (int r = 0; r < 4; r++) k[r] = lane * 4 + r < n ? pts[off + lane * 4 + r] : MM_EMPTY;
      mm_wave_bitonic<4>(k, lane);
#pragma unroll
      for (int r = 0; r < 4; r++) if (lane * 4 + r < n) pts[off + lane * 4 + r] = k[r];
    } else {
      uint64_t k[8];
#pragma unroll
      for (int r = 0; r < 8; r++) k[r] = lane * 8 + r < n ? pts[off + lane * 8 + r] : MM_EMPTY;
      mm_wave_bitonic<8>(k, lane);
#pragma unroll
      for (int r = 0; r < 8; r++) if (lane * 8 + r < n) pts[off + lane * 8 + r] = k[r];
    }
  }
}

// 65..LDSCAP points (power of two): one 256-thread workgroup per fragment, bitonic sort staged in LDS
#define MM_SORT_LDSCAP 4096
// ids (may be null): a 16-bit payload that travels with its key (windowed mode: the seed of every point)
__global__ void __launch_bounds__(256)
k_sort_points_block(const unsigned int* __restrict__ nListDev, const int64_t* __restrict__ ptOff, uint64_t* __restrict__ pts, const int32_t* __restrict__ list, uint16_t* __restrict__ ids) {
  __shared__ uint64_t sk[MM_SORT_LDSCAP];
  __shared__ uint16_t si[MM_SORT_LDSCAP];
  const int nList = (int)*nListDev;
  for (int li = blockIdx.x; li < nList; li += gridDim.x) {
    const int f = list[li];
    const int64_t off = ptOff[2 * f]; const int n = (int)ptOff[2 * f + 1];
    for (int i = threadIdx.x; i < n; i += 256) { sk[i] = pts[off + i]; if (ids) si[i] = ids[off + i]; }
    __syncthreads();
    for (int k = 2; k <= n; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < n; i += 256) {
          const int p = i ^ j;
          if (p > i) {
            const uint64_t a = sk[i], b = sk[p];
            if ((a > b) == ((i & k) == 0)) { sk[i] = b; sk[p] = a; if (ids) { const uint16_t t = si[i]; si[i] = si[p]; si[p] = t; } }
          }
        }
        __syncthreads();
      }
    for (int i = threadIdx.x; i < n; i += 256) { pts[off + i] = sk[i]; if (ids) ids[off + i] = si[i]; }
    __syncthreads();
  }
}

// > LDSCAP points: one 1024-thread workgroup per fragment, bitonic sort in global memory (rare: very repetitive seeds)
__global__ void __launch_bounds__(1024)
k_sort_points_global(const unsigned int* __restrict__ nListDev, const int64_t* __restrict__ ptOff, uint64_t* __restrict__ pts, const int32_t* __restrict__ list, uint16_t* __restrict__ ids) {
  const int nList = (int)*nListDev;
  for (int li = blockIdx.x; li < nList; li += gridDim.x) {
    const int f = list[li];
    const int64_t off = ptOff[2 * f]; const int64_t n = ptOff[2 * f + 1];
    uint64_t* a = pts + off;
    uint16_t* d = ids ? ids + off : nullptr;
    for (int64_t k = 2; k <= n; k <<= 1)
      for (int64_t j = k >> 1; j > 0; j >>= 1) {
        for (int64_t i = threadIdx.x; i < n; i += 1024) {
          const int64_t p = i ^ j;
          if (p > i) { const uint64_t x = a[i], y = a[p]; if ((x > y) == ((i & k) == 0)) { a[i] = y; a[p] = x; if (d) { const uint16_t t = d[i]; d[i] = d[p]; d[p] = t; } } }
        }
        __threadfence_block();
        __syncthreads();
      }
  }
}

// splits the queued fragments into the block / global sorter lists (wave-aggregated cursors)
__global__ void k_classify_sort(int nList, const unsigned long long* __restrict__ nDev, const int32_t* __restrict__ list, const int64_t* __restrict__ ptOff,
                                int32_t* __restrict__ listB, int32_t* __restrict__ listC, unsigned int* __restrict__ cnt /* [0] B, [1] C */, int minB /* lists longer than this go to the LDS sorter */) {
  if (nDev) nList = (int)*nDev;
  for (int base = blockIdx.x * blockDim.x; base < nList; base += gridDim.x * blockDim.x) {
    const int li = base + threadIdx.x;
    int f = -1; int64_t n = 0;
    if (li < nList) { f = list[li]; n = ptOff[2 * f + 1]; }
    const bool isC = n > MM_SORT_LDSCAP, isB = !isC && n > minB;
    const uint64_t mB = __ballot(isB), mC = __ballot(isC);
    unsigned int bB = 0, bC = 0;
    if (mB && mm_lane() == (uint32_t)__builtin_ctzll(mB)) bB = atomicAdd(&cnt[0], (unsigned int)__popcll(mB));
    if (mC && mm_lane() == (uint32_t)__builtin_ctzll(mC)) bC = atomicAdd(&cnt[1], (unsigned int)__popcll(mC));
    if (mB) bB = __shfl(bB, __builtin_ctzll(mB));
    if (mC) bC = __shfl(bC, __builtin_ctzll(mC));
    if (isB) listB[bB + mm_popc_below(mB)] = f;
    if (isC) listC[bC + mm_popc_below(mC)] = f;
  }
}

// ---------------------------------------------------------------------------------------------
// k_l1_sweep: one thread per queued fragment over its sorted points (windowLen == 0, i.e. split mode).
// Literal two-pointer restatement of computeMap.hpp:948-1115 on packed keys:
//   key>>1 == (seqId<<32 | pos)  so  "trail <= lead in (seqId,pos)"  is one 64-bit compare.
// Emits the joined candidates of each reference group (skip_prefix) in the reference's order.
// ---------------------------------------------------------------------------------------------
struct L1Emit {
  mm_l1_candidate* out; int frag; int count; bool write;
  bool have; mm_l1_candidate pend;
  __device__ __forceinline__ void run(int seqId, int start, int end, int isize, int clusterLen, bool& firstOfGroup) {
    // join with the previous candidate of the same computeL1CandidateRegions call when close (computeMap.hpp:1102-1115)
    if (have && !firstOfGroup && seqId == pend.seqId && !(start > pend.rangeEndPos + clusterLen)) {
      pend.rangeEndPos = end; pend.intersectionSize = isize > pend.intersectionSize ? isize : pend.intersectionSize;
    } else {
      flush();
      pend.frag = frag; pend.seqId = seqId; pend.rangeStartPos = start; pend.rangeEndPos = end; pend.intersectionSize = isize; have = true;
    }
    firstOfGroup = false;
  }
  mm_l1_candidate b0, b1;                             // the first two candidates of a counting pass: most fragments need no second pass
  __device__ __forceinline__ void flush() {
    if (have) {
      if (write) out[count] = pend;
      else if (count == 0) b0 = pend;
      else if (count == 1) b1 = pend;
      count++; have = false;
    }
  }
};

__device__ void l1_sweep_fragment(const uint64_t* __restrict__ p, int nPts, int sketchSizeQ, int minHits0, const int32_t* __restrict__ cutoffs,
                                  int nCutoffs, int sParam, int segLength, MapFlags fl, const int32_t* __restrict__ refGroup, L1Emit& em) {
  int b = 0;
  while (b < nPts) {
    int e = nPts;
    if (fl.skipPrefix) {
      const int g = refGroup[(int)(p[b] >> 33)];
      e = b; while (e < nPts && refGroup[(int)(p[e] >> 33)] == g) e++;
    }
    int minHits = minHits0;
    bool go = true;
    if (fl.hg) {                                                 // pass 1: best overlap (computeMap.hpp:948-999)
      int overlap = 0, best = 0, trail = b, lead = b;
      while (lead < e) {
        const uint64_t lk = p[lead] >> 1;
        while (trail < e && (p[trail] >> 1) <= lk) { if (!(p[trail] & 1ull)) overlap--; trail++; }
        const uint32_t cur = (uint32_t)lk;
        while (lead < e && (uint32_t)(p[lead] >> 1) == cur) { if (p[lead] & 1ull) overlap++; lead++; }
        best = overlap > best ? overlap : best;
      }
      if (best < minHits) go = false;
      else {
        const double div = (double)sParam / 1000.0 > 1.0 ? (double)sParam / 1000.0 : 1.0;
        int ci = (int)((double)(best < sketchSizeQ ? best : sketchSizeQ) / div);
        if (ci >= nCutoffs) ci = nCutoffs - 1;
        const int cut = cutoffs[ci];
        minHits = cut > minHits ? cut : minHits;
      }
    }
    if (go) {                                                    // pass 2: runs (computeMap.hpp:1009-1098)
      bool firstOfGroup = true, inRun = false;
      int rSeq = 0, rStart = 0, rEnd = 0, rSize = 0;
      int overlap = 0, trail = b, lead = b;
      int prevSeq = 0, prevPos = 0;
      int curSeq = (int)(p[b] >> 33), curPos = (int)(uint32_t)(p[b] >> 1);
      while (lead < e) {
        const int prevOverlap = overlap;
        const uint64_t lk = p[lead] >> 1;
        while (trail < e && (p[trail] >> 1) <= lk) { if (!(p[trail] & 1ull)) overlap--; trail++; }
        if ((int)(uint32_t)lk != curPos) { prevSeq = curSeq; prevPos = curPos; curSeq = (int)(lk >> 32); curPos = (int)(uint32_t)lk; }
        while (lead < e && (int)(uint32_t)(p[lead] >> 1) == curPos) { if (p[lead] & 1ull) overlap++; lead++; }
        if (prevOverlap >= minHits) {
          if (inRun && rSeq != prevSeq) { em.run(rSeq, rStart, rEnd, rSize, segLength, firstOfGroup); inRun = false; }
          if (!inRun) { rStart = prevPos; rEnd = prevPos; rSeq = prevSeq; rSize = prevOverlap; inRun = true; }
          else { rSize = prevOverlap > rSize ? prevOverlap : rSize; rEnd = prevPos; }
        } else {
          if (inRun) em.run(rSeq, rStart, rEnd, rSize, segLength, firstOfGroup);
          inRun = false;
        }
      }
      if (inRun) em.run(rSeq, rStart, rEnd, rSize, segLength, firstOfGroup);
    }
    em.flush();
    b = e;
  }
}

// ---------------------------------------------------------------------------------------------
// k_l1_stream: the L1 stage of a queued fragment by one WAVE, streaming its sorted points from HBM 64 at a time (coalesced) -- any
// number of points.  Same formulation as mm_l1_fused: the overlap count after a position group is the running sum of +1 (OPEN) / -1
// (CLOSE) up to its last point; pass 1 finds the best count (computeMap.hpp:948-999), pass 2 folds the groups whose count reaches
// minimumHits -- the last group never does, :1024-1098 -- into runs per contig and joins runs closer than segLength (:1102-1115).  The
// fold is sequential, as in the reference, but wave-uniform and only over the groups that start a run or end one (everything between
// two of those extends the run: a masked wave maximum).  A position group that spans two contigs or minimumHits <= 0 leaves the fragment to the literal k_l1_sweep
// (list `lit`).  Fragments with many points are the rule in repeat families (segmental duplications: every locus x copies), where the
// one-thread-per-fragment kernel costs 50-170 ns per fragment (profiles/r03j_repeat_probe.txt).
// ---------------------------------------------------------------------------------------------
#define MM_STREAM_BUF 64            // candidates kept in LDS by the counting pass (more: the writing pass runs again)
struct L1RunS { int32_t seq, start, end, isize; };
__global__ void __launch_bounds__(256)
k_l1_stream(int nList, const int32_t* __restrict__ list, const int64_t* __restrict__ ptOff, const uint64_t* __restrict__ pts,
            mm_frag_stats* __restrict__ stats, const int32_t* __restrict__ minHitsTab, const int32_t* __restrict__ cutoffs, int nCutoffs,
            int sParam, int segLength, int hg, mm_l1_candidate* __restrict__ l1, unsigned long long l1Cap, int64_t* __restrict__ l1Off,
            int32_t* __restrict__ lit, unsigned int* __restrict__ litCount, unsigned long long* __restrict__ counters /* [2] l1 cursor, [3] overflow */,
            const unsigned long long* __restrict__ nDev, const int32_t* __restrict__ ptKept /* points at the head of the sorted list (k_gather_points / k_filter_points) */) {
  __shared__ L1RunS bufAll[4][MM_STREAM_BUF];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  L1RunS* buf = bufAll[wave];
  if (nDev) nList = (int)*nDev;
  auto one = [&](const int f) {
  const int nPts = ptKept[f], S = stats[f].sketchSize;
  if (nPts <= 0 || S <= 0) { if (lane == 0) { stats[f].nL1 = 0; l1Off[f] = 0; } return; }
  const uint64_t* p = pts + ptOff[2 * f];
  int minHits = minHitsTab[S];

  // one chunk of 64 points: key, overlap count after the point, whether it ends a position group
  auto chunk = [&](int i0, int carry, uint64_t& k, int& run, bool& gLast, bool& mixed, int& sum) {
    const int idx = i0 + lane;
    const bool valid = idx < nPts;
    k = valid ? p[idx] : MM_EMPTY;
    const uint64_t prv = (valid && idx > 0) ? p[idx - 1] : MM_EMPTY;
    const uint64_t nxt = (idx + 1 < nPts) ? p[idx + 1] : MM_EMPTY;
    mixed = valid && prv != MM_EMPTY && (uint32_t)(prv >> 1) == (uint32_t)(k >> 1) && (prv >> 33) != (k >> 33);
    const int delta = valid ? ((k & 1ull) ? 1 : -1) : 0;
    run = carry + mm_wave_excl_scan(delta) + delta;
    gLast = valid && (nxt == MM_EMPTY || (uint32_t)(nxt >> 1) != (uint32_t)(k >> 1));
    sum = mm_wave_sum(delta);
  };

  // ---- pass 1: best overlap count, number of groups, the mixed-group test ----
  int best = 0, G = 0; bool anyMixed = false;
  {
    int carry = 0;
    for (int i0 = 0; i0 < nPts; i0 += 64) {
      uint64_t k; int run, sum; bool gLast, mixed;
      chunk(i0, carry, k, run, gLast, mixed, sum);
      if (gLast && run > best) best = run;
      G += (int)__popcll(__ballot(gLast));
      anyMixed = anyMixed || __ballot(mixed) != 0;
      carry += sum;
    }
    best = mm_wave_max(best);
  }
  bool go = G > 0;
  if (go && hg) {                                                 // computeMap.hpp:984-998
    if (best < minHits) go = false;
    else {
      const double div = (double)sParam / 1000.0 > 1.0 ? (double)sParam / 1000.0 : 1.0;
      int ci = (int)((double)(best < S ? best : S) / div);
      if (ci >= nCutoffs) ci = nCutoffs - 1;
      const int cut = cutoffs[ci];
      minHits = cut > minHits ? cut : minHits;
    }
  }
  if (anyMixed || minHits <= 0) {                                 // the literal kernel's business
    if (lane == 0) lit[atomicAdd(litCount, 1u)] = f;
    return;
  }

  // ---- pass 2: runs of flagged groups -> joined candidates; counted (the first MM_STREAM_BUF kept in LDS), then written ----
  int total = 0; long long base = 0;
  for (int pass = 0; pass < 2; pass++) {
    const bool write = pass == 1;
    int count = 0;
    bool have = false; L1RunS pend{0, 0, 0, 0};
    auto flush = [&]() {
      if (have) {
        if (lane == 0) {
          if (write) { mm_l1_candidate o; o.frag = f; o.seqId = pend.seq; o.rangeStartPos = pend.start; o.rangeEndPos = pend.end; o.intersectionSize = pend.isize; l1[base + count] = o; }
          else if (count < MM_STREAM_BUF) buf[count] = pend;
        }
        count++; have = false;
      }
    };
    auto emit = [&](int seq, int start, int end, int isize) {      // L1Emit::run for a single reference group
      if (have && seq == pend.seq && !(start > pend.end + segLength)) { pend.end = end; pend.isize = isize > pend.isize ? isize : pend.isize; }
      else { flush(); pend.seq = seq; pend.start = start; pend.end = end; pend.isize = isize; have = true; }
    };
    if (go) {
      bool inRun = false; int rSeq = 0, rStart = 0, rEnd = 0, rSize = 0;
      int carry = 0, gBase = 0;
      bool prevFlagC = false; int prevSeqC = 0;                   // the last group of the chunks so far: flagged?, its contig
      for (int i0 = 0; i0 < nPts; i0 += 64) {
        uint64_t k; int run, sum; bool gLast, mixed;
        chunk(i0, carry, k, run, gLast, mixed, sum);
        carry += sum;
        const uint64_t GM = __ballot(gLast);
        if (!GM) continue;
        const int gidx = gBase + (int)mm_popc_below(GM);
        gBase += (int)__popcll(GM);
        const bool flag = gLast && gidx < G - 1 && run >= minHits;
        const uint64_t FM = __ballot(flag);
        const int gseq = (int)(k >> 33), gpos = (int)(uint32_t)(k >> 1);
        const int lastG = 63 - (int)__builtin_clzll(GM);
        const bool lastFlag = (FM >> lastG) & 1ull; const int lastSeq = __builtin_amdgcn_readlane(gseq, lastG);
        if (FM || prevFlagC) {
          // the group before this lane's: in the chunk, or the carried one
          const uint64_t below = GM & ((1ull << lane) - 1ull);
          const int pl = below ? 63 - (int)__builtin_clzll(below) : lane;
          const int sseq = __shfl(gseq, pl);
          const bool pf = below ? (((FM >> pl) & 1ull) != 0) : prevFlagC;
          const int ps = below ? sseq : prevSeqC;
          // a flagged group continues the run of a flagged predecessor on the same contig; the groups that matter are the others:
          // flagged ones that start a run, and unflagged ones right behind a flagged one, which end it.  Between two of those every
          // flagged group just extends the run: its end is the last one's position, its size the largest count.
          const uint64_t startM = __ballot(flag && !(pf && ps == gseq));
          const uint64_t breakM = __ballot(gLast && !flag && pf);
          auto extend = [&](int lo, int hi) {                       // flagged groups of the lanes [lo, hi) join the open run
            if (lo >= 64) return;
            const uint64_t m = FM & ~((1ull << lo) - 1ull) & (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull));
            if (!m) return;
            rEnd = __builtin_amdgcn_readlane(gpos, 63 - (int)__builtin_clzll(m));
            const int mx = mm_wave_max((flag && lane >= lo && lane < hi) ? run : 0);
            rSize = mx > rSize ? mx : rSize;
          };
          uint64_t ev = startM | breakM;
          int segLo = 0;
          while (ev) {
            const int bpos = (int)__builtin_ctzll(ev);
            ev &= ev - 1ull;
            if (inRun) extend(segLo, bpos);
            if ((breakM >> bpos) & 1ull) { if (inRun) emit(rSeq, rStart, rEnd, rSize); inRun = false; }
            else {
              if (inRun) emit(rSeq, rStart, rEnd, rSize);
              rSeq = __builtin_amdgcn_readlane(gseq, bpos); rStart = __builtin_amdgcn_readlane(gpos, bpos); rEnd = rStart;
              rSize = __builtin_amdgcn_readlane(run, bpos); inRun = true;
            }
            segLo = bpos + 1;
          }
          if (inRun) extend(segLo, 64);
        }
        prevFlagC = lastFlag; prevSeqC = lastSeq;
      }
      if (inRun) emit(rSeq, rStart, rEnd, rSize);
      flush();
    }
    if (!write) {
      total = count;
      if (total > 0) {
        long long b0 = 0;
        if (lane == 0) b0 = (long long)atomicAdd(&counters[2], (unsigned long long)total);
        base = ((long long)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)b0);
        if ((unsigned long long)base + (unsigned long long)total > l1Cap) { if (lane == 0) atomicOr(&counters[3], 1ull); total = 0; }
      }
      if (total > 0 && total <= MM_STREAM_BUF) {                   // the counting pass kept them all
        __threadfence_block();
        if (lane < total) { const L1RunS r = buf[lane]; mm_l1_candidate o; o.frag = f; o.seqId = r.seq; o.rangeStartPos = r.start; o.rangeEndPos = r.end; o.intersectionSize = r.isize; l1[base + lane] = o; }
        break;
      }
      if (total == 0) break;
    }
  }
  if (lane == 0) { stats[f].nL1 = total; l1Off[f] = total > 0 ? base : 0; }
  };
  for (int li = blockIdx.x * 4 + wave; li < nList; li += gridDim.x * 4) { one(list[li]); __threadfence_block(); }
}

__global__ void __launch_bounds__(256)
k_l1_sweep(int nList, const int32_t* __restrict__ list, const int64_t* __restrict__ ptOff, const uint64_t* __restrict__ pts,
           mm_frag_stats* __restrict__ stats,
           const int32_t* __restrict__ minHitsTab, const int32_t* __restrict__ cutoffs, int nCutoffs, int sParam, int segLength,
           MapFlags fl, const int32_t* __restrict__ refGroup, mm_l1_candidate* __restrict__ l1, unsigned long long l1Cap,
           int64_t* __restrict__ l1Off, unsigned long long* __restrict__ counters /* [2] l1 cursor, [3] overflow */,
           const unsigned int* __restrict__ nListDev /* non-null: the list's length lives on the device (what k_l1_stream left over) */,
           const unsigned long long* __restrict__ nDev64 /* non-null: the same as a 64-bit counter (the hand-over count of k_lookup_l1) */,
           const int32_t* __restrict__ ptKept) {
  if (nListDev) nList = (int)*nListDev;
  if (nDev64) nList = (int)*nDev64;
  for (int li = blockIdx.x * blockDim.x + threadIdx.x; li < nList; li += gridDim.x * blockDim.x) {
  const int f = list[li];
  const int nPts = ptKept[f], S = stats[f].sketchSize;
  int nOut = 0; long long base = 0;
  if (nPts > 0 && S > 0) {
    const uint64_t* p = pts + ptOff[2 * f];
    const int minHits0 = minHitsTab[S];
    L1Emit em; em.out = nullptr; em.frag = f; em.count = 0; em.write = false; em.have = false;
    l1_sweep_fragment(p, nPts, S, minHits0, cutoffs, nCutoffs, sParam, segLength, fl, refGroup, em);
    nOut = em.count;
    if (nOut > 0) {
      base = (long long)atomicAdd(&counters[2], (unsigned long long)nOut);
      if ((unsigned long long)base + nOut > l1Cap) { atomicOr(&counters[3], 1ull); nOut = 0; }
      else if (nOut <= 2) { l1[base] = em.b0; if (nOut == 2) l1[base + 1] = em.b1; }
      else {
        L1Emit ew; ew.out = l1 + base; ew.frag = f; ew.count = 0; ew.write = true; ew.have = false;
        l1_sweep_fragment(p, nPts, S, minHits0, cutoffs, nCutoffs, sParam, segLength, fl, refGroup, ew);
      }
    }
  }
  stats[f].nL1 = nOut;
  l1Off[f] = base;
  }
}


// ---------------------------------------------------------------------------------------------
// k_l1_window: computeL1CandidateRegions for fragments LONGER than segLength (--noSplit: Q.len > segLength, windowLen =
// Q.len - segLength != 0, computeMap.hpp:933), literally: one thread per fragment over its sorted points, the trailing pointer windowLen
// behind the leading one, and a count of open windows per seed (hash_to_freq, :948) -- a seed adds to the overlap only while its count
// goes 0 -> 1 and leaves it only when it returns to 0.  `ids` numbers the seeds of a fragment (mm_gather_points); `freq` is this
// thread's slice of a zero-initialised scratch array.  Handles windowLen == 0 as well (a batch may mix short and long reads).
// ---------------------------------------------------------------------------------------------
__device__ void l1_window_fragment(const uint64_t* __restrict__ p, const uint16_t* __restrict__ ids, int nPts, int W, int32_t* __restrict__ freq, int nFreq,
                                   int sketchSizeQ, int minHits0, const int32_t* __restrict__ cutoffs, int nCutoffs, int sParam, int segLength, MapFlags fl,
                                   const int32_t* __restrict__ refGroup, L1Emit& em) {
  auto seqOf = [&](int i) { return (int)(p[i] >> 33); };
  auto posOf = [&](int i) { return (int)(uint32_t)(p[i] >> 1); };
  auto behind = [&](int t, int l) { const int st = seqOf(t), sl = seqOf(l); return (st == sl && posOf(t) <= posOf(l) - W) || st < sl; };   // :952-954
  auto closeAt = [&](int t, int& overlap) { if (!(p[t] & 1ull)) { const int id = ids[t]; if (W != 0) freq[id]--; if (W == 0 || freq[id] == 0) overlap--; } };
  auto openAt = [&](int l, int& overlap) { if (p[l] & 1ull) { const int id = ids[l]; if (W == 0 || freq[id] == 0) overlap++; if (W != 0) freq[id]++; } };
  int b = 0;
  while (b < nPts) {
    int e = nPts;
    if (fl.skipPrefix) {
      const int g = refGroup[seqOf(b)];
      e = b; while (e < nPts && refGroup[seqOf(e)] == g) e++;
    }
    int minHits = minHits0;
    bool go = true;
    if (fl.hg) {                                                 // pass 1: best overlap (:948-999)
      for (int i = 0; i < nFreq; i++) freq[i] = 0;
      int overlap = 0, best = 0, trail = b, lead = b;
      while (lead < e) {
        while (trail < e && behind(trail, lead)) { closeAt(trail, overlap); trail++; }
        const int cur = posOf(lead);
        while (lead < e && posOf(lead) == cur) { openAt(lead, overlap); lead++; }
        best = overlap > best ? overlap : best;
      }
      if (best < minHits) go = false;
      else {
        const double div = (double)sParam / 1000.0 > 1.0 ? (double)sParam / 1000.0 : 1.0;
        int ci = (int)((double)(best < sketchSizeQ ? best : sketchSizeQ) / div);
        if (ci >= nCutoffs) ci = nCutoffs - 1;
        const int cut = cutoffs[ci];
        minHits = cut > minHits ? cut : minHits;
      }
    }
    if (go) {                                                    // pass 2 (:1001-1098); hash_to_freq.clear() first (:1001-1003)
      for (int i = 0; i < nFreq; i++) freq[i] = 0;
      bool firstOfGroup = true, inRun = false;
      int rSeq = 0, rStart = 0, rEnd = 0, rSize = 0;
      int overlap = 0, trail = b, lead = b;
      int prevSeq = 0, prevPos = 0;
      int curSeq = seqOf(b), curPos = posOf(b);
      while (lead < e) {
        const int prevOverlap = overlap;
        while (trail < e && behind(trail, lead)) { closeAt(trail, overlap); trail++; }
        if (posOf(lead) != curPos) { prevSeq = curSeq; prevPos = curPos; curSeq = seqOf(lead); curPos = posOf(lead); }
        while (lead < e && posOf(lead) == curPos) { openAt(lead, overlap); lead++; }
        if (prevOverlap >= minHits) {
          if (inRun && rSeq != prevSeq) { em.run(rSeq, rStart, rEnd, rSize, segLength, firstOfGroup); inRun = false; }
          if (!inRun) { rStart = prevPos - W; rEnd = prevPos - W; rSeq = prevSeq; rSize = prevOverlap; inRun = true; }
          else { rSize = prevOverlap > rSize ? prevOverlap : rSize; rEnd = prevPos - W; }
        } else {
          if (inRun) em.run(rSeq, rStart, rEnd, rSize, segLength, firstOfGroup);
          inRun = false;
        }
      }
      if (inRun) em.run(rSeq, rStart, rEnd, rSize, segLength, firstOfGroup);
    }
    em.flush();
    b = e;
  }
}

__global__ void __launch_bounds__(64)
k_l1_window(int nList, const int32_t* __restrict__ list, const DFrag* __restrict__ frags, const int64_t* __restrict__ ptOff, const uint64_t* __restrict__ pts,
            const uint16_t* __restrict__ ptIds, mm_frag_stats* __restrict__ stats, const int32_t* __restrict__ minHitsTab, const int32_t* __restrict__ cutoffs,
            int nCutoffs, int sParam, int segLength, MapFlags fl, const int32_t* __restrict__ refGroup, int32_t* __restrict__ freqAll, int nFreq,
            mm_l1_candidate* __restrict__ l1, unsigned long long l1Cap, int64_t* __restrict__ l1Off, unsigned long long* __restrict__ counters /* [2] l1 cursor, [3] overflow */) {
  const int li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= nList) return;
  const int f = list[li];
  const int nPts = stats[f].nPoints, S = stats[f].sketchSize;
  int nOut = 0; long long base = 0;
  if (nPts > 0 && S > 0) {
    const uint64_t* p = pts + ptOff[2 * f];
    const uint16_t* ids = ptIds + ptOff[2 * f];
    int W = frags[f].len - segLength; if (W < 0) W = 0;
    int32_t* freq = freqAll + (size_t)li * nFreq;
    const int minHits0 = minHitsTab[S];
    L1Emit em; em.out = nullptr; em.frag = f; em.count = 0; em.write = false; em.have = false;
    l1_window_fragment(p, ids, nPts, W, freq, nFreq, S, minHits0, cutoffs, nCutoffs, sParam, segLength, fl, refGroup, em);
    nOut = em.count;
    if (nOut > 0) {
      base = (long long)atomicAdd(&counters[2], (unsigned long long)nOut);
      if ((unsigned long long)base + nOut > l1Cap) { atomicOr(&counters[3], 1ull); nOut = 0; }
      else if (nOut <= 2) { l1[base] = em.b0; if (nOut == 2) l1[base + 1] = em.b1; }
      else {
        L1Emit ew; ew.out = l1 + base; ew.frag = f; ew.count = 0; ew.write = true; ew.have = false;
        l1_window_fragment(p, ids, nPts, W, freq, nFreq, S, minHits0, cutoffs, nCutoffs, sParam, segLength, fl, refGroup, ew);
      }
    }
  }
  stats[f].nL1 = nOut;
  l1Off[f] = base;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
#define MM_SYNC(c) do { MM_HIP(c, hipStreamSynchronize((c)->stream)); (c)->nSyncs++; } while (0)

// One pass of seed lookup + L1 + L2 + selection over the resident sketches.
//   steady == false: every stage is sized from the counts of the one before it, read back from the device (and grown + retried when a
//                    buffer overflows): five to seven host synchronisations per pass.
//   steady == true : the previous pass of this context has sized every buffer; everything is launched against those capacities with
//                    the counts left on the device (kernels over lists take their lengths from there, kernels over candidates cover the
//                    buffers' capacity), and one block of counters is read back at the end: ONE synchronisation per pass.  A pass that
//                    outgrew a buffer has its overflow flag set there and is redone the sized way (MM_PASS_REDO).
static int map_pass(mm_ctx* c, const bool steady) {
  const int nF = (int)c->nFrags, s = c->P.sketchSize;
  const DeviceIndex& I = c->idx;
  MapFlags fl{(c->P.flags & MM_FLAG_HG_FILTER) ? 1 : 0, (c->P.flags & MM_FLAG_SKIP_SELF) ? 1 : 0,
              (c->P.flags & MM_FLAG_SKIP_PREFIX) ? 1 : 0, (c->P.flags & MM_FLAG_LOWER_TRIANGULAR) ? 1 : 0};
  // fragments longer than segLength (--noSplit): windowLen != 0 -- every fragment takes the literal path with its points (and their
  // seeds' numbers) in HBM: k_l1_window here, k_l2_window in mm_l2.hip
  const bool windowed = c->windowed;
  const bool allSlow = c->keepPoints || fl.skipPrefix || windowed;
  const size_t cF = mm_frag_cap(c, (size_t)nF);
  if (c->ptsCap == 0) c->ptsCap = allSlow ? cF * 128 + 4096 : cF * 8 + 65536;
  if (c->l1Cap < cF * 2 + 1024) c->l1Cap = cF * 2 + 1024;
  DevBuf& listB = c->dListB; DevBuf& listC = c->dListC;
  MM_HIP(c, listB.ensure(cF * 4 + 16)); MM_HIP(c, listC.ensure(cF * 4 + 16)); MM_HIP(c, c->dBigList.ensure(cF * 4 + 16));
  // fragments with more interval points than k_lookup_l1 sorts go through k_lookup_mid first (MM_NO_MID=1: straight to the HBM path, the A/B switch)
  static const bool noMid = getenv("MM_NO_MID") != nullptr;
  const bool useMid = !allSlow && !noMid && s <= MM_MID_MAXSKETCH;
  if (useMid) MM_HIP(c, c->dMidList.ensure(cF * 4 + 16));
  unsigned long long hMid = 0;
  MM_HIP(c, c->dL1Regions.ensure(sizeof(L1Regions)));
  int rc = MM_OK;
  unsigned long long hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long* cnt = c->dCounters.as<unsigned long long>() + 8;   // [8..15]; [0..7] belong to the sketch launcher
  unsigned long long* cnt2 = c->dCounters.as<unsigned long long>() + 32; // [32..]: [0] candidate mappings [1] their buffer overflowed
  const unsigned long long* nBigDev = steady ? cnt + 7 : nullptr;

  const SeedTable seedTab{I.htSlots.as<HtSlot>(), (uint64_t)(I.htCap - 1), I.filter.as<uint64_t>(), (uint64_t)I.filterMask, I.tagged ? I.htTags.as<uint8_t>() : (const uint8_t*)nullptr};
  unsigned long long hcur[MM_L1_REGIONS * MM_L1_CURSOR_STRIDE] = {0};   // a steady-state pass never reads the cursors back
  unsigned long long regionCap = 0;
  for (int attempt = 0; attempt < 10; attempt++) {          // grow-and-retry on capacity overflow (points or L1 candidates)
    regionCap = (c->l1Cap + MM_L1_REGIONS - 1) / MM_L1_REGIONS + 64;
    MM_HIP(c, c->dPts.ensure(c->ptsCap * 8 + 64));
    MM_HIP(c, c->dPtIds.ensure(windowed ? c->ptsCap * 2 + 64 : 64));
    MM_HIP(c, c->dL1.ensure(regionCap * MM_L1_REGIONS * sizeof(mm_l1_candidate) + 64));
    MM_HIP(c, c->dL1b.ensure(regionCap * MM_L1_REGIONS * sizeof(mm_l1_candidate) + 64));
    MM_HIP(c, c->dL1Cursors.ensure(sizeof hcur));
    if (attempt == 0) MM_HIP(c, hipMemcpyAsync(c->dCounters.as<unsigned long long>() + 48, c->dCounters.p, 8, hipMemcpyDeviceToDevice, c->stream));   // the sketch launcher's hard-list length ([0], 32 bits) survives the reset below
    MM_HIP(c, hipMemsetAsync(c->dCounters.p, 0, 256, c->stream));
    MM_HIP(c, hipMemsetAsync(cnt2, 0, 64, c->stream));
    MM_HIP(c, hipMemsetAsync(c->dL1Cursors.p, 0, sizeof hcur, c->stream));
    {
      KernelTimer t(c, MM_K_LOOKUP);
      // the fused path holds a fragment's interval points in LDS + registers: up to 128 of them for the default sketch, 256 / 512 for
      // larger ones (points come in proportion to the sketch: s = 310 averages ~176 per fragment with a long tail)
      int fuse = s > 256 ? 512 : s > 160 ? 256 : 128;
      if (const char* e = getenv("MM_FUSE_MAXPTS")) { const int v = atoi(e); fuse = v >= 512 ? 512 : v >= 256 ? 256 : 128; }
      auto kern = I.tagged ? (fuse == 512 ? k_lookup_l1<512, true> : fuse == 256 ? k_lookup_l1<256, true> : k_lookup_l1<128, true>)
                           : (fuse == 512 ? k_lookup_l1<512, false> : fuse == 256 ? k_lookup_l1<256, false> : k_lookup_l1<128, false>);
      hipLaunchKernelGGL(kern, dim3((nF + MM_LOOKUP_WPB - 1) / MM_LOOKUP_WPB), dim3(MM_LOOKUP_WPB * 64), 0, c->stream, nF, s, c->dFrags.as<DFrag>(),
                         c->dSkHash.as<uint64_t>(), c->dSkStrand.as<int8_t>(), c->dSkCount.as<uint32_t>(), seedTab, I.ptKeys.as<uint64_t>(),
                         c->dReadSelf.as<int32_t>(), c->seqCounterBase, fl, windowed ? 2 : (c->keepPoints ? 1 : 0),
                         c->dQHash.as<uint64_t>(), c->dQStrand.as<int8_t>(),
                         c->dStats.as<mm_frag_stats>(), c->dPtOff.as<int64_t>(), (unsigned long long)c->ptsCap,
                         c->dMinHits.as<int32_t>(), c->dCutoffs.as<int32_t>(), (int)c->nCutoffs, c->P.segLength,
                         c->dL1.as<mm_l1_candidate>(), regionCap, c->dL1Cursors.as<unsigned long long>(), c->dL1Off.as<int64_t>(),
                         c->dBigList.as<int32_t>(), useMid ? c->dMidList.as<int32_t>() : (int32_t*)nullptr, cnt);
      MM_HIP(c, hipGetLastError());
    }
    if (useMid) {
      // its list's length stays on the device (cnt[16]); the grid is what the last sized pass saw plus a half (the kernel walks the list
      // with whatever grid it gets), every fragment when nothing is known yet
      KernelTimer t(c, MM_K_SORT);
      const size_t guess = c->midKnown ? c->prevMid + c->prevMid / 2 + 256 : (size_t)nF;
      const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(guess, (size_t)nF));
      auto mk = I.tagged ? k_lookup_mid<true> : k_lookup_mid<false>;
      hipLaunchKernelGGL(mk, dim3(grid), dim3(64), 0, c->stream, 0, cnt + 16, c->dMidList.as<int32_t>(), s, c->dFrags.as<DFrag>(), c->dSkHash.as<uint64_t>(),
                         c->dSkCount.as<uint32_t>(), seedTab, I.ptKeys.as<uint64_t>(), c->dReadSelf.as<int32_t>(), c->seqCounterBase, fl, c->dStats.as<mm_frag_stats>(),
                         c->dPtOff.as<int64_t>(), (unsigned long long)c->ptsCap, c->dMinHits.as<int32_t>(), c->dCutoffs.as<int32_t>(), (int)c->nCutoffs, c->P.segLength,
                         c->dL1.as<mm_l1_candidate>(), regionCap, c->dL1Cursors.as<unsigned long long>(), c->dL1Off.as<int64_t>(), c->dBigList.as<int32_t>(), cnt);
      MM_HIP(c, hipGetLastError());
    }
    if (steady) break;                                        // overflow flags ([1], [3]) are looked at when the pass is over
    MM_HIP(c, hipMemcpyAsync(hc, cnt, 64, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(&hMid, cnt + 16, 8, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(hcur, c->dL1Cursors.p, sizeof hcur, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(c->hPass + 16, c->dCounters.as<unsigned long long>() + 48, 8, hipMemcpyDeviceToHost, c->stream));
    MM_SYNC(c);
    c->lastHard = (size_t)(c->hPass[16] & 0xffffffffull);
    if (hc[1]) { const size_t need = mm_scaled(c, (size_t)hc[0], 16); c->ptsCap = need + need / 8 + 4096; continue; }
    if (hc[3]) {                                              // some region overflowed: size for the largest one seen
      unsigned long long mx = 0;
      for (int r = 0; r < MM_L1_REGIONS; r++) mx = std::max(mx, hcur[(size_t)r * MM_L1_CURSOR_STRIDE]);
      mx = (unsigned long long)mm_scaled(c, (size_t)mx, sizeof(mm_l1_candidate) * MM_L1_REGIONS);
      c->l1Cap = (size_t)(mx + mx / 4 + 256) * MM_L1_REGIONS; continue;
    }
    break;
  }
  if (hc[1]) { c->err = "interval-point buffer overflow"; return MM_ERR_CAPACITY; }
  if (hc[3]) { c->err = "L1 candidate buffer overflow"; return MM_ERR_CAPACITY; }
  {
    // the regions' prefix positions and the number of fused candidates (cnt[2]) are computed on the device; the sized pass knows them
    // from the cursors it has read
    unsigned long long tot = 0;
    for (int r = 0; r < MM_L1_REGIONS; r++) tot += hcur[(size_t)r * MM_L1_CURSOR_STRIDE];
    hc[2] = steady ? 0 : tot;
    if (steady || tot) {
      KernelTimer t(c, MM_K_L1);
      hipLaunchKernelGGL(k_l1_regions, dim3(1), dim3(64), 0, c->stream, c->dL1Cursors.as<unsigned long long>(), regionCap, c->dL1Regions.as<L1Regions>(), cnt);
      hipLaunchKernelGGL(k_l1_compact, dim3(64, MM_L1_REGIONS), dim3(256), 0, c->stream, c->dL1.as<mm_l1_candidate>(), c->dL1b.as<mm_l1_candidate>(), regionCap, c->dL1Regions.as<L1Regions>());
      hipLaunchKernelGGL(k_l1_fix_offsets, dim3((nF + 255) / 256), dim3(256), 0, c->stream, nF, c->dStats.as<mm_frag_stats>(), c->dL1Off.as<int64_t>(), regionCap, c->dL1Regions.as<L1Regions>());
      MM_HIP(c, hipGetLastError());
    }
    std::swap(c->dL1, c->dL1b);                               // dL1 is the dense buffer from here on
  }
  size_t denseCap = c->dL1.bytes / sizeof(mm_l1_candidate) - 4;   // room behind the fused candidates for the sweep path's
  const int nBig = (int)hc[7];
  if (getenv("MM_DEBUG")) {
    if (steady) fprintf(stderr, "[mm] lookup+L1: %d fragments, steady-state pass (the counts stay on the device)\n", nF);
    else fprintf(stderr, "[mm] lookup+L1: %d fragments, %llu through k_lookup_mid, %d to the sort+sweep path, %llu fused candidates\n", nF, hMid, nBig, hc[2]);
  }
  if (steady || nBig > 0) {
    unsigned int* cls = (unsigned int*)(c->dCounters.as<unsigned long long>() + 16);   // [16] two 32-bit class counters
    // grids over the queued fragments: exact when their number is known, otherwise sized by what the previous pass saw (the kernels
    // walk the list with whatever grid they get)
    const int nB = steady ? (int)std::min<size_t>(c->prevBig + c->prevBig / 2 + 1024, (size_t)nF) : nBig;
    const unsigned gWave = (unsigned)((nB + 3) / 4), gThread = (unsigned)((nB + 255) / 256);
    {
      KernelTimer t(c, MM_K_SORT);
      uint16_t* sortIds = windowed ? c->dPtIds.as<uint16_t>() : (uint16_t*)nullptr;
      {
        auto gk = I.tagged ? k_gather_points<true> : k_gather_points<false>;
        hipLaunchKernelGGL(gk, dim3(gWave), dim3(256), 0, c->stream, nBig, c->dBigList.as<int32_t>(), s, c->dFrags.as<DFrag>(), c->dSkHash.as<uint64_t>(),
                           c->dSkCount.as<uint32_t>(), seedTab, I.ptKeys.as<uint64_t>(), I.refGroup.as<int32_t>(),
                           c->dReadGroup.as<int32_t>(), c->dReadSelf.as<int32_t>(), c->seqCounterBase, fl, c->dStats.as<mm_frag_stats>(), c->dPtOff.as<int64_t>(),
                           c->dPts.as<uint64_t>(), sortIds, c->dPtKept.as<int32_t>(), nBigDev);
        MM_HIP(c, hipGetLastError());
        // split mode, plain L1: the points that cannot reach minimumHits go before the sort (k_filter_points; MM_NO_POINT_FILTER=1 is the A/B switch)
        static const bool noFilter = getenv("MM_NO_POINT_FILTER") != nullptr;
        if (!windowed && (!c->keepPoints || c->keepFiltered) && !fl.skipPrefix && !noFilter) {
          hipLaunchKernelGGL(k_filter_points, dim3(gWave), dim3(256), 0, c->stream, nBig, c->dBigList.as<int32_t>(), c->dStats.as<mm_frag_stats>(), c->dMinHits.as<int32_t>(),
                             c->dPtOff.as<int64_t>(), c->dPts.as<uint64_t>(), c->dPtKept.as<int32_t>(), nBigDev);
          MM_HIP(c, hipGetLastError());
        }
      }
      if (!windowed) hipLaunchKernelGGL(k_sort_points_wave, dim3(gWave), dim3(256), 0, c->stream, nBig, nBigDev, c->dBigList.as<int32_t>(), c->dPtOff.as<int64_t>(), c->dPts.as<uint64_t>());
      hipLaunchKernelGGL(k_classify_sort, dim3(gThread), dim3(256), 0, c->stream, nBig, nBigDev, c->dBigList.as<int32_t>(), c->dPtOff.as<int64_t>(),
                         listB.as<int32_t>(), listC.as<int32_t>(), cls, windowed ? 1 : MM_SORT_WAVECAP);
      MM_HIP(c, hipGetLastError());
      unsigned int hcls[2] = {1024u, 16u};                    // steady: fixed grids over the two lists (their lengths stay on the device)
      if (!steady) {
        MM_HIP(c, hipMemcpyAsync(hcls, cls, 8, hipMemcpyDeviceToHost, c->stream));
        MM_SYNC(c);
        if (getenv("MM_DEBUG")) fprintf(stderr, "[mm] point path: %d queued fragments, after the filter %u lists of more than %d points (LDS sorter), %u beyond the LDS sorter\n", nBig, hcls[0], MM_SORT_WAVECAP, hcls[1]);
      }
      if (hcls[0]) hipLaunchKernelGGL(k_sort_points_block, dim3(hcls[0]), dim3(256), 0, c->stream, cls, c->dPtOff.as<int64_t>(), c->dPts.as<uint64_t>(), listB.as<int32_t>(), sortIds);
      if (hcls[1]) hipLaunchKernelGGL(k_sort_points_global, dim3(hcls[1]), dim3(1024), 0, c->stream, cls + 1, c->dPtOff.as<int64_t>(), c->dPts.as<uint64_t>(), listC.as<int32_t>(), sortIds);
      MM_HIP(c, hipGetLastError());
    }
    for (int attempt = 0; attempt < 8; attempt++) {
      // candidates of the fused path sit at [0, fusedL1); the sweep appends behind them, so a retry only rewinds to fusedL1
      const unsigned long long fusedL1 = hc[2];
      if (!steady) {
        MM_HIP(c, hipMemcpyAsync(cnt + 2, &fusedL1, 8, hipMemcpyHostToDevice, c->stream));
        MM_HIP(c, hipMemsetAsync(cnt + 3, 0, 8, c->stream));
      }
      {
        KernelTimer t(c, MM_K_L1);
        // a wave per fragment streams the sorted points; what it cannot take (a position group across two contigs, minimumHits 0) and
        // every fragment under -Y reference groups goes to the literal one-thread-per-fragment kernel
        const bool stream = !fl.skipPrefix && !getenv("MM_L1_LITERAL") && !windowed;
        const int32_t* sweepList = c->dBigList.as<int32_t>(); const unsigned int* sweepCount = nullptr;
        if (windowed) {
          const int nFreq = s > 256 ? s : 256;                                     // seeds are numbered by their index in the raw sketch
          MM_HIP(c, c->dWinFreq.ensure((size_t)nBig * nFreq * 4 + 64));
          hipLaunchKernelGGL(k_l1_window, dim3((nBig + 63) / 64), dim3(64), 0, c->stream, nBig, c->dBigList.as<int32_t>(), c->dFrags.as<DFrag>(), c->dPtOff.as<int64_t>(),
                             c->dPts.as<uint64_t>(), c->dPtIds.as<uint16_t>(), c->dStats.as<mm_frag_stats>(), c->dMinHits.as<int32_t>(), c->dCutoffs.as<int32_t>(),
                             (int)c->nCutoffs, s, c->P.segLength, fl, I.refGroup.as<int32_t>(), c->dWinFreq.as<int32_t>(), nFreq, c->dL1.as<mm_l1_candidate>(),
                             (unsigned long long)denseCap, c->dL1Off.as<int64_t>(), cnt);
          MM_HIP(c, hipGetLastError());
        } else {
        unsigned int* lit = cls + 2;                                               // [17]: the list k_l1_stream leaves over (cls itself still feeds the sorters in flight)
        if (stream) {
          MM_HIP(c, hipMemsetAsync(lit, 0, 8, c->stream));
          hipLaunchKernelGGL(k_l1_stream, dim3(gWave), dim3(256), 0, c->stream, nBig, c->dBigList.as<int32_t>(), c->dPtOff.as<int64_t>(),
                             c->dPts.as<uint64_t>(), c->dStats.as<mm_frag_stats>(), c->dMinHits.as<int32_t>(), c->dCutoffs.as<int32_t>(),
                             (int)c->nCutoffs, s, c->P.segLength, fl.hg, c->dL1.as<mm_l1_candidate>(), (unsigned long long)denseCap,
                             c->dL1Off.as<int64_t>(), listB.as<int32_t>(), lit, cnt, nBigDev, c->dPtKept.as<int32_t>());
          MM_HIP(c, hipGetLastError());
          sweepList = listB.as<int32_t>(); sweepCount = lit;
        }
        hipLaunchKernelGGL(k_l1_sweep, dim3(stream && steady ? 64u : gThread), dim3(256), 0, c->stream, nBig, sweepList, c->dPtOff.as<int64_t>(),
                           c->dPts.as<uint64_t>(), c->dStats.as<mm_frag_stats>(), c->dMinHits.as<int32_t>(), c->dCutoffs.as<int32_t>(),
                           (int)c->nCutoffs, s, c->P.segLength, fl, I.refGroup.as<int32_t>(), c->dL1.as<mm_l1_candidate>(),
                           (unsigned long long)denseCap, c->dL1Off.as<int64_t>(), cnt, sweepCount, sweepCount ? (const unsigned long long*)nullptr : nBigDev, c->dPtKept.as<int32_t>());
        MM_HIP(c, hipGetLastError());
        }
      }
      if (steady) break;
      unsigned long long h2[2];
      MM_HIP(c, hipMemcpyAsync(h2, cnt + 2, 16, hipMemcpyDeviceToHost, c->stream));
      MM_SYNC(c);
      if (h2[1]) {
        // grow, keeping the fused candidates already in the buffer
        const size_t newCap = (size_t)h2[0] + (size_t)h2[0] / 8 + 1024;
        DevBuf nb; MM_HIP(c, nb.ensure(newCap * sizeof(mm_l1_candidate) + 64));
        if (fusedL1) MM_HIP(c, hipMemcpyAsync(nb.p, c->dL1.p, (size_t)fusedL1 * sizeof(mm_l1_candidate), hipMemcpyDeviceToDevice, c->stream));
        MM_SYNC(c);
        c->dL1.release(); c->dL1 = nb; denseCap = newCap;
        if (c->l1Cap < newCap) c->l1Cap = newCap;                  // the next pass sizes both candidate buffers for it: no retry, no reallocation
        hc[3] = 1; continue;
      }
      hc[2] = h2[0]; hc[3] = 0;
      break;
    }
    if (hc[3]) { c->err = "L1 candidate buffer overflow"; return MM_ERR_CAPACITY; }
  }
  if (!steady) {
    c->nL1 = (size_t)hc[2];
    c->prevBig = c->lastBig = (size_t)nBig;
    c->prevMid = c->lastMid = (size_t)hMid; c->midKnown = useMid;
    if (c->nL1 == 0) { c->nL2 = 0; return rc; }
  }
  if (steady) {
    // candidates the buffers of this pass hold: what the previous (sized) pass left of candidate-indexed staging, and the dense L1 buffer itself
    const unsigned long long cap = (unsigned long long)std::min(c->candCap, denseCap);
    hipLaunchKernelGGL(k_l1_gate, dim3(1), dim3(1), 0, c->stream, cnt, cap);
    MM_HIP(c, hipGetLastError());
  }
  rc = mm_launch_l2(c, cnt, steady);
  if (rc != MM_OK) return rc;
  rc = mm_launch_select(c, steady);
  if (rc != MM_OK) return rc;
  if (steady) {
    // the pass is over: one block of counters comes back
    unsigned long long* h = c->hPass;
    MM_HIP(c, hipMemcpyAsync(h, cnt, 64, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(h + 8, cnt2, 64, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(h + 16, c->dCounters.as<unsigned long long>() + 48, 8, hipMemcpyDeviceToHost, c->stream));
    MM_SYNC(c);
    c->lastHard = (size_t)(h[16] & 0xffffffffull);
    if (h[1] || h[3] || h[5] || (h[6] & ~0ull) || h[9]) return MM_PASS_REDO;   // some buffer was too small for this batch: the sized pass grows it
    c->nL1 = (size_t)h[2]; c->nL2 = (size_t)h[4]; c->nMappings = c->haveReplayTables ? (size_t)h[8] : 0;
    c->lastOps = (size_t)h[10]; c->lastBig = c->prevBig;          // (the queue's length was overwritten by the L2 stage's list counter: the sized pass's stands in)
  }
  return MM_OK;
}

int mm_launch_map(mm_ctx* c) {
  const int nF = (int)c->nFrags, s = c->P.sketchSize;
  const size_t cF = mm_frag_cap(c, (size_t)nF);
  MM_HIP(c, c->dQHash.ensure(cF * s * 8 + 64)); MM_HIP(c, c->dQStrand.ensure(cF * s + 64));
  MM_HIP(c, c->dStats.ensure(cF * sizeof(mm_frag_stats) + 64));
  MM_HIP(c, c->dPtOff.ensure(cF * 16 + 64)); MM_HIP(c, c->dL1Off.ensure(cF * 8 + 64)); MM_HIP(c, c->dPtKept.ensure(cF * 4 + 64));
  MM_HIP(c, c->dCounters.ensure(512));
  if (!c->hPass) { MM_HIP(c, hipHostMalloc((void**)&c->hPass, 256, hipHostMallocDefault)); }
  c->nL1 = c->nL2 = 0; c->nMappings = 0; c->nSyncs = 0; c->lastSteady = false;
  if (nF == 0) return MM_OK;
  const bool allSlow = c->keepPoints || (c->P.flags & MM_FLAG_SKIP_PREFIX) || c->windowed || c->P.sketchSize > MM_LDS_MAX_SKETCH;
  static const bool noSteady = getenv("MM_NO_STEADY") != nullptr;
  // (a batch with a tenth more fragments than the one the buffers were sized for would only fail and be redone: it is sized right away)
  if (c->steadyOk && c->steadyFails < 3 && !allSlow && !noSteady && (size_t)nF <= c->sizedFrags + c->sizedFrags / 10) {
    const int rc = map_pass(c, true);
    if (rc == MM_OK) { c->lastSteady = true; c->steadyFails = 0; return MM_OK; }
    if (rc != MM_PASS_REDO) return rc;
    c->steadyOk = false; c->steadyFails++; c->nRedone++;                        // three redone passes in a row: this context's batches keep outgrowing what the one before left
  }
  const int rc = map_pass(c, false);
  c->sizedFrags = mm_frag_cap(c, (size_t)nF);
  c->steadyOk = rc == MM_OK && !allSlow && c->nL1 > 0 && c->l2Chunks == 1;   // (a batch whose L2 streams go through in chunks needs the host between them)
  return rc;
}

// ---------------------------------------------------------------------------------------------
// downloads
// ---------------------------------------------------------------------------------------------
extern "C" {

int mm_results_download(mm_ctx* c, mm_frag_stats* stats, mm_l1_candidate* l1, mm_l2_locus* l2) {
  if (!c->mapped) { c->err = "mm_results_download: nothing mapped"; return MM_ERR_STATE; }
  MM_HIP(c, hipSetDevice(c->device));
  const size_t nF = c->nFrags;
  std::vector<int64_t> l1off(nF);
  std::vector<mm_frag_stats> st(nF);
  if (nF) {
    MM_HIP(c, hipMemcpyAsync(st.data(), c->dStats.p, nF * sizeof(mm_frag_stats), hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(l1off.data(), c->dL1Off.p, nF * 8, hipMemcpyDeviceToHost, c->stream));
  }
  std::vector<mm_l1_candidate> h1(c->nL1); std::vector<mm_l2_locus> h2(c->nL2);
  if (c->nL1) MM_HIP(c, hipMemcpyAsync(h1.data(), c->dL1.p, c->nL1 * sizeof(mm_l1_candidate), hipMemcpyDeviceToHost, c->stream));
  if (c->nL2) MM_HIP(c, hipMemcpyAsync(h2.data(), c->dL2.p, c->nL2 * sizeof(mm_l2_locus), hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  if (stats && nF) std::memcpy(stats, st.data(), nF * sizeof(mm_frag_stats));
  // fragment-major order: candidates of fragment f are contiguous on the device at l1off[f]
  std::vector<int64_t> newBase(nF, 0);
  size_t w = 0;
  for (size_t f = 0; f < nF; f++) {
    newBase[f] = (int64_t)w;
    for (int i = 0; i < st[f].nL1; i++) { if (l1) l1[w] = h1[(size_t)l1off[f] + i]; w++; }
  }
  if (l2 && c->nL2) {
    // device order keeps every candidate's loci contiguous and in emission order; a stable sort by (frag, cand) finishes the job
    std::vector<uint32_t> ord(c->nL2);
    std::iota(ord.begin(), ord.end(), 0u);
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
      if (h2[a].frag != h2[b].frag) return h2[a].frag < h2[b].frag;
      return h2[a].cand < h2[b].cand; });
    for (size_t i = 0; i < c->nL2; i++) { mm_l2_locus o = h2[ord[i]]; o.cand = (int32_t)(newBase[o.frag] + o.cand); l2[i] = o; }
  }
  return MM_OK;
}

int mm_query_sketch_download(mm_ctx* c, mm_minmer* out) {
  if (!c->mapped) { c->err = "mm_query_sketch_download: nothing mapped"; return MM_ERR_STATE; }
  MM_HIP(c, hipSetDevice(c->device));
  const size_t nF = c->nFrags, s = (size_t)c->P.sketchSize;
  if (!nF) return MM_OK;
  std::vector<uint64_t> h(nF * s); std::vector<int8_t> st(nF * s); std::vector<mm_frag_stats> fs(nF);
  MM_HIP(c, hipMemcpyAsync(h.data(), c->dQHash.p, nF * s * 8, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(st.data(), c->dQStrand.p, nF * s, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(fs.data(), c->dStats.p, nF * sizeof(mm_frag_stats), hipMemcpyDeviceToHost, c->stream));
  // fragments that lost no seed were not copied to dQHash/dQStrand: their sketch is the raw one (k_lookup_l1)
  std::vector<uint64_t> rh(nF * s); std::vector<int8_t> rst(nF * s);
  MM_HIP(c, hipMemcpyAsync(rh.data(), c->dSkHash.p, nF * s * 8, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(rst.data(), c->dSkStrand.p, nF * s, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  for (size_t f = 0; f < nF; f++) {
    const bool raw = fs[f].rawSketchSize == fs[f].sketchSize;
    for (int r = 0; r < fs[f].sketchSize; r++) {
      const size_t o = f * s + r;
      out[o] = mm_minmer{raw ? rh[o] : h[o], 0, 0, c->hFrags[f].readId + c->seqCounterBase, (int16_t)(raw ? rst[o] : st[o]), 0};
    }
  }
  return MM_OK;
}

int mm_points_download(mm_ctx* c, size_t frag, mm_interval_point* out, size_t cap, size_t* n) {
  if (!c->mapped || frag >= c->nFrags) { c->err = "mm_points_download: bad state / fragment"; return MM_ERR_STATE; }
  if (!c->keepPoints && !(c->P.flags & MM_FLAG_SKIP_PREFIX)) { c->err = "mm_points_download: interval points are not kept in HBM (mm_set_option MM_OPT_KEEP_POINTS)"; return MM_ERR_STATE; }
  MM_HIP(c, hipSetDevice(c->device));
  int64_t po[2]; mm_frag_stats fs;
  MM_HIP(c, hipMemcpy(po, c->dPtOff.as<int64_t>() + 2 * frag, 16, hipMemcpyDeviceToHost));
  MM_HIP(c, hipMemcpy(&fs, c->dStats.as<mm_frag_stats>() + frag, sizeof fs, hipMemcpyDeviceToHost));
  size_t np = (size_t)fs.nPoints;
  if (c->keepFiltered) {                                            // the head of the sorted list k_filter_points left (dropped points sort behind it)
    int32_t kept = 0;
    MM_HIP(c, hipMemcpy(&kept, c->dPtKept.as<int32_t>() + frag, 4, hipMemcpyDeviceToHost));
    if (po[1] > 0) np = (size_t)kept;
  }
  if (n) *n = np;
  if (np > cap) { c->err = "mm_points_download: capacity"; return MM_ERR_ARG; }
  std::vector<uint64_t> k(np);
  if (np) MM_HIP(c, hipMemcpy(k.data(), c->dPts.as<uint64_t>() + po[0], np * 8, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < np; i++) {
    std::memset(&out[i], 0, sizeof out[i]);
    out[i].seqId = (int32_t)(k[i] >> 33); out[i].pos = (int32_t)(uint32_t)(k[i] >> 1); out[i].side = (k[i] & 1ull) ? 1 : -1;
  }
  return MM_OK;
}


}  // extern "C"

// mashmap_amd/csrc/mm_sketch.hip -- a1 (pack) and a4 (query-fragment sketch) kernels.
//
//   k_pack2bit          makeUpperCaseAndValidDNA            src/map/include/commonFunc.hpp:97
//   k_sketch_fast / k_sketch_hard  CommonFunc::sketchSequence          src/map/include/commonFunc.hpp:183-288
//
// One workgroup per query fragment.  Integer-ALU bound (2 x MurmurHash3_x64_128 per base); the
// packed input is only 0.25 B/bp (+0.125 B/bp N mask).  See DESIGN.md for the roofline terms.
#include "mm_internal.h"
#include "mm_device.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// k_pack2bit: ASCII -> 2 bit/base + 1 bit/base N mask.  One thread per 32 output bases.
// Reads are laid out at 32-base aligned offsets so code words (16 bases) and mask words (32 bases)
// of different reads never share a word.
// ---------------------------------------------------------------------------------------------
__global__ void k_pack2bit(const uint8_t* __restrict__ ascii, const int64_t* __restrict__ srcOff,
                           const int64_t* __restrict__ packOff, const int32_t* __restrict__ readLen, int nReads,
                           int64_t nChunks, uint32_t* __restrict__ bases2, uint32_t* __restrict__ nmask,
                           uint32_t* __restrict__ readHasN) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nChunks; c += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b0 = c * 32;
    int lo = 0, hi = nReads;                        // last read with packOff <= b0
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (packOff[mid] <= b0) lo = mid; else hi = mid; }
    const int r = lo;
    const int64_t local = b0 - packOff[r];
    const int len = readLen[r];
    uint32_t wlo = 0, whi = 0, nm = 0;
    if (local < len) {
      const uint8_t* src = ascii + srcOff[r] + local;
      const int cnt = (len - local) < 32 ? (int)(len - local) : 32;
      for (int i = 0; i < cnt; i++) {
        const uint32_t ch = src[i] & 0xDFu;         // a-z -> A-Z (only letters can land on A/C/G/T)
        const bool ok = (ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T');
        const uint32_t code = ok ? (((ch >> 1) ^ (ch >> 2)) & 3u) : 0u;   // A0 C1 G2 T3
        if (i < 16) wlo |= code << (2 * i); else whi |= code << (2 * (i - 16));
        nm |= (ok ? 0u : 1u) << i;
      }
    }
    bases2[2 * c] = wlo; bases2[2 * c + 1] = whi;
    nmask[c] = nm;
    if (nm) atomicOr(&readHasN[r], 1u);
  }
}

// ---------------------------------------------------------------------------------------------
// LDS table that de-duplicates the survivors of the threshold filter AND orders them.
//   The home slot is a monotone function of the hash (hashes below the cut T are spread over
//   [0, HT)), collisions probe linearly WITHOUT wrap-around into `pad` spare slots.  Then
//     * every key of a cluster (maximal run of occupied slots) has its home inside the cluster,
//     * keys of an earlier cluster have a strictly smaller home, hence a strictly smaller hash,
//   so rank(key) = occupied slots before its cluster + keys of its own cluster that are smaller:
//   the table scan replaces compaction + sort.
// ---------------------------------------------------------------------------------------------
#define MM_SK_PAD 64               // spill slots behind the ordered tables (no wrap-around): fast kernel
#define MM_SK_PADH 256             // hard kernel
#define MM_SK_DUPCAP 256          // repeat occurrences a fast-kernel fragment may defer (more: the hard kernel takes the fragment)
#define MM_SK_GUARD 4             // always-empty slots on either side of the table: the ranking reads windows of that many neighbours
struct SkTable {                  // the hard kernel's table: every occurrence is folded in with atomics as it is hashed
  uint64_t* key; int32_t* first; int32_t* last; int32_t* sum;
  uint32_t* counters;            // [0] distinct keys  [1] overflow (load limit passed, spill area full)  [2] occupied slots
  uint64_t* occW; uint32_t* occP; // occupancy bit words and their exclusive prefix counts
  uint32_t nSlots, maxLoad, M; int sh;

  // slots for hashes in [0, T): x = top 32 bits of h after normalising T to bit 63, home = x * M >> 32 with
  // M <= 2^32 * HT / (x_max + 1)  (any smaller M keeps home < HT and monotone; the float estimate is shaded down)
  __device__ __forceinline__ void set_cut(uint64_t T, uint32_t HT) {
    sh = (T == MM_HASH_MAX) ? 0 : __clzll((long long)T);
    const uint64_t t32 = ((T << sh) >> 32) + 1ull;
    M = (uint32_t)((float)HT * 4294967296.0f / (float)t32 * 0.99999f);
  }
  __device__ __forceinline__ uint32_t home(uint64_t h) const { return __umulhi((uint32_t)((h << sh) >> 32), M); }

  // the number of distinct keys is kept in counters[0] and the load limit flagged as it is passed: the kernel inserts straight from
  // the hash loop and must notice a flooded table early
  __device__ __forceinline__ void insert(uint64_t h, int pos, int st) {
    uint32_t slot = home(h);
    for (uint32_t probes = 1;; probes++) {
      // a flooded table (cut still too high) is abandoned early; looked at every 16th probe only, the volatile generic load is slow
      if ((probes & 15u) == 0 && ((volatile uint32_t*)counters)[1]) return;
      const unsigned long long prev = atomicCAS((unsigned long long*)&key[slot], (unsigned long long)MM_HASH_MAX,
                                                (unsigned long long)h);
      if (prev == MM_HASH_MAX) {
        if (atomicAdd(&counters[0], 1u) >= maxLoad) atomicOr(&counters[1], 1u);
      }
      if (prev == MM_HASH_MAX || prev == h) {
        atomicMin(&first[slot], pos); atomicMax(&last[slot], pos); atomicAdd(&sum[slot], st);
        return;
      }
      if (++slot >= nSlots) { atomicOr(&counters[1], 1u); return; }
    }
  }
};

// words of the fast kernel's position queue: the per-wave queues, but at least the table's load limit (the memory lists the occupied slots later)
__host__ __device__ static inline int sketch_fast_queue_total(int QC, int nWaves, int HT) { const int a = QC * nWaves, b = HT * 5 / 8 + 1; return ((a > b ? a : b) + 3) & ~3; }

// OR of every K-wide window: bit j of the result = any of bits j..j+K-1 of x
template <int K>
__device__ __forceinline__ uint64_t mm_window_or(uint64_t x) {
  constexpr int P = K >= 32 ? 32 : K >= 16 ? 16 : K >= 8 ? 8 : K >= 4 ? 4 : K >= 2 ? 2 : 1;
  uint64_t w = x;
#pragma unroll
  for (int d = 1; d < P; d <<= 1) w |= w >> d;
  return w | (w >> (K - P));
}

// ---------------------------------------------------------------------------------------------
// k_sketch_hard<K>: the fragments k_sketch_fast gave up on -- tandem repeats, low complexity, N-rich -- exact for any input.
//   Survivors go straight from the hash loop into a larger table (duplicates collapse there, however many), and the cut T is
//   moved until s <= distinct <= load limit: first in proportion to the distinct count the previous cut produced (the fast
//   kernel leaves its own count in skCount[f] as the first hint), then by bisection once a cut has flooded the table.
// ---------------------------------------------------------------------------------------------
#define MM_SK_HINT_OVERFLOW 0xFFFFFFFFu   // skCount[f] of a listed fragment: a queue / table / duplicate list of the fast kernel overflowed
// SPILL: the first / last / strand-sum arrays of the table live in HBM scratch (`spill`: 3 x NS int32 per workgroup) instead of LDS --
// sketches beyond ~1500 entries, whose key table alone fills most of a CU's 160 KB (the reference's --dense derives sketchSize =
// 0.02 (1 + (1 - pi) / 0.05) (segLength - k): 1998 for --pi 80 -s 20000, parseCmdArgs.hpp:626-630).  Same atomics, slower memory; exact all the same.
template <int K, bool SPILL>
__device__ __forceinline__ void
mm_sketch_hard(unsigned char* smem, const int f, const uint4* __restrict__ gTabs, const uint32_t* __restrict__ bases2, const uint32_t* __restrict__ nmask,
               const DFrag* __restrict__ frags, const uint32_t* __restrict__ readHasN, int s, int wantFast, int HT, int PAD, int32_t* __restrict__ spill,
               const bool stream /* the fragment does not fit the LDS: its words are read from global memory as they are needed */,
               uint64_t* __restrict__ skHash, int2* __restrict__ skPos, int8_t* __restrict__ skStrand, uint32_t* __restrict__ skCount) {
  const DFrag fr = frags[f];
  const int len = fr.len;
  const int n = len - K + 1;                        // k-mer positions
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (n <= 0) { if (tid == 0) skCount[f] = 0; return; }
  const bool hasN = readHasN[fr.readId] != 0;
  const uint32_t hint = skCount[f];                 // distinct survivors under the fast kernel's cut, or MM_SK_HINT_OVERFLOW

  // ---- LDS carve (every offset a multiple of 16) ----
  const int nW = stream ? 0 : (len + 15) / 16 + 3;  // code words incl. 2 words of run-off for the last strip
  const int nM = stream ? 0 : (len + 31) / 32 + 2;
  const int NS = HT + PAD;                          // table slots (multiple of 64)
  const int nOcc = NS >> 6;
  size_t off = 0;
  using Tabs = typename MMTabsFor<K>::type;
  Tabs* tabs = (Tabs*)(smem + off); off += sizeof(Tabs);
  static_assert(sizeof(Tabs) % 16 == 0, "tables are copied in 16-byte pieces");
  for (int i = tid; i < (int)(sizeof(Tabs) / 16); i += nthr) ((uint4*)tabs)[i] = gTabs[i];   // visible after the first __syncthreads() below
  uint32_t* sW = (uint32_t*)(smem + off); off += (((size_t)nW * 4 + 15) / 16) * 16;
  uint32_t* sM = (uint32_t*)(smem + off); off += (((size_t)nM * 4 + 15) / 16) * 16;
  SkTable tab;
  tab.key = (uint64_t*)(smem + off) + MM_SK_GUARD; off += (size_t)(NS + 2 * MM_SK_GUARD) * 8;   // MM_SK_GUARD always-empty slots on either side: the ranking reads windows
  tab.occW = (uint64_t*)(smem + off); off += (size_t)nOcc * 8;
  if constexpr (SPILL) {
    int32_t* g = spill + (size_t)blockIdx.x * 3 * (size_t)NS;
    tab.first = g; tab.last = g + NS; tab.sum = g + 2 * (size_t)NS;
  } else {
    tab.first = (int32_t*)(smem + off); off += (size_t)NS * 4;
    tab.last = (int32_t*)(smem + off); off += (size_t)NS * 4;
    tab.sum = (int32_t*)(smem + off); off += (size_t)NS * 4;
  }
  tab.occP = (uint32_t*)(smem + off); off += (((size_t)nOcc * 4 + 15) / 16) * 16;
  tab.counters = (uint32_t*)(smem + off); off += 16;
  tab.nSlots = (uint32_t)NS; tab.maxLoad = (uint32_t)HT * 5u / 8u;

  // ---- stage the fragment, re-aligned so that LDS word j holds bases 16j..16j+15 ----
  const int64_t gw0 = fr.base >> 4; const int gsh = (int)(fr.base & 15) * 2;
  const int64_t gm0 = fr.base >> 5; const int gmsh = (int)(fr.base & 31);
  {
    for (int j = tid; j < nW; j += nthr) {
      const uint32_t a = bases2[gw0 + j], b = bases2[gw0 + j + 1];
      sW[j] = gsh ? __builtin_amdgcn_alignbit(b, a, gsh) : a;
    }
    if (hasN) {
      for (int j = tid; j < nM; j += nthr) {
        const uint32_t a = nmask[gm0 + j], b = nmask[gm0 + j + 1];
        sM[j] = gmsh ? __builtin_amdgcn_alignbit(b, a, gmsh) : a;
      }
    }
  }
  // word j of the re-aligned fragment / of its N mask: from the LDS copy, or -- a fragment longer than the LDS (a whole contig under
  // --noSplit) -- straight from the packed batch, realigned on the fly (neighbouring strips share their words in the caches)
  auto wordAt = [&](int j) -> uint32_t {
    if (!stream) return sW[j];
    const uint32_t a = bases2[gw0 + j], b = bases2[gw0 + j + 1];
    return gsh ? __builtin_amdgcn_alignbit(b, a, gsh) : a;
  };
  auto maskAt = [&](int j) -> uint32_t {
    if (!stream) return sM[j];
    const uint32_t a = nmask[gm0 + j], b = nmask[gm0 + j + 1];
    return gmsh ? __builtin_amdgcn_alignbit(b, a, gmsh) : a;
  };

  // Any cut gives the exact sketch as long as >= s distinct hashes survive it (checked below), so float estimates are enough.
  // `scaled(T, D)`: the cut under which 1.6 s distinct hashes are expected when T produced D (distinct counts grow with the cut
  // in proportion for all but pathological inputs; the loop below corrects those), at least twice T.
  auto scaled = [&](uint64_t T, uint32_t D) -> uint64_t {
    float r = 1.6f * (float)s / (float)(D ? D : 1u);
    if (r < 2.0f) r = 2.0f;
    const float t = (float)T * r;
    return t >= 1.8e19f ? MM_HASH_MAX : (uint64_t)t;
  };
  uint64_t T = ((uint32_t)wantFast >= (uint32_t)n) ? MM_HASH_MAX : (uint64_t)((float)wantFast / (float)(2 * n) * 18446744073709551616.0f);   // the fast kernel's cut
  if (hint != MM_SK_HINT_OVERFLOW && T != MM_HASH_MAX) T = scaled(T, hint);
  uint64_t lo = 0, hi = MM_HASH_MAX; bool hiInf = true;   // bisection state (uniform across the block)
  const int nStrips = (n + 15) >> 4;
  uint32_t D = 0;

  for (int attempt = 0;; attempt++) {
    tab.set_cut(T, (uint32_t)HT);
    for (int i = tid; i < NS; i += nthr) { tab.key[i] = MM_HASH_MAX; tab.first[i] = 0x7fffffff; tab.last[i] = -1; tab.sum[i] = 0; }
    if (tid < MM_SK_GUARD) { tab.key[-1 - tid] = MM_HASH_MAX; tab.key[NS + tid] = MM_HASH_MAX; }
    if (tid < 4) tab.counters[tid] = 0;
    if constexpr (SPILL) __threadfence();                 // the spilled arrays' initial values, before other threads' atomics reach them
    __syncthreads();

    // ---- hash both strands of every k-mer ----
    const bool allPass = (T == MM_HASH_MAX);
    for (int strip = tid; strip < nStrips; strip += nthr) {
      // bit j: position strip*16+j exists and its k-mer holds no N
      const int rem = n - strip * 16;
      uint32_t ok = rem >= 16 ? 0xFFFFu : ((1u << rem) - 1u);
      if (hasN) {
        const uint64_t m64 = (uint64_t)maskAt(strip >> 1) | ((uint64_t)maskAt((strip >> 1) + 1) << 32);
        if constexpr (MMWideK<K>::value) {
          const uint64_t h64 = (uint64_t)maskAt((strip >> 1) + 2) | ((uint64_t)maskAt((strip >> 1) + 3) << 32);
          const int sh = (strip & 1) * 16;
          ok &= ~mm_window_or_wide<K>(sh ? ((m64 >> sh) | (h64 << (64 - sh))) : m64, sh ? (h64 >> sh) : h64);
        } else ok &= ~(uint32_t)mm_window_or<K>(m64 >> ((strip & 1) * 16));
      }
      auto onPos = [&](int j, uint64_t hf, uint64_t hr) {
        const int pos = strip * 16 + j;
        const uint64_t h = hf < hr ? hf : hr;
        bool pass = false;                          // nested ifs: the compiler keeps the three tests as exec masks
        if ((ok & (1u << j)) != 0) { if (hf != hr) { if (allPass || h < T) pass = true; } }
        if (pass) tab.insert(h, pos, hf < hr ? 1 : -1);
      };
      if constexpr (MMWideK<K>::value) {
        uint32_t ww[MMWideK<K>::NW];
#pragma unroll
        for (int i = 0; i < MMWideK<K>::NW; i++) ww[i] = wordAt(strip + i);
        mm_strip_hashes_wide<K>(ww, *tabs, onPos);
      } else mm_strip_hashes<K>(wordAt(strip), wordAt(strip + 1), wordAt(strip + 2), *tabs, onPos);
    }
    if constexpr (SPILL) __threadfence();                 // the atomics on the spilled arrays, before the ranking threads read them
    __syncthreads();

    // ---- occupancy words, their prefix counts (wave 0: one DPP scan per 64 words), number of distinct keys ----
    for (int base = 0; base < NS; base += nthr) {
      const int slot = base + tid;                  // NS and nthr are multiples of 64: a wave is in or out as a whole
      if (slot < NS) {
        const uint64_t m = __ballot(tab.key[slot] != MM_HASH_MAX);
        if (mm_lane() == 0) tab.occW[slot >> 6] = m;
      }
    }
    __syncthreads();
    if (tid < 64) {
      int carry = 0;
      for (int w0 = 0; w0 < nOcc; w0 += 64) {
        const int w = w0 + tid;
        const int v = w < nOcc ? (int)__popcll(tab.occW[w]) : 0;
        const int ex = mm_wave_excl_scan(v);
        if (w < nOcc) tab.occP[w] = (uint32_t)(carry + ex);
        carry += mm_wave_sum(v);
      }
      if (tid == 0) tab.counters[2] = (uint32_t)carry;
    }
    __syncthreads();

    D = tab.counters[2];
    const bool overflow = tab.counters[1] != 0;
    if (overflow) { hi = T; hiInf = false; }
    else if (D < (uint32_t)s && T != MM_HASH_MAX) { lo = T; }
    else break;
    T = hiInf ? scaled(T, D) : lo + (hi - lo) / 2;
    __syncthreads();
    if (attempt > 200) break;                       // cannot happen (bisection on 64 bits); keeps the loop bounded
  }

  // ---- rank every key inside its cluster; emit the s smallest, ascending (commonFunc.hpp:278-286) ----
  for (int slot = tid; slot < NS; slot += nthr) {
    const uint64_t k = tab.key[slot];
    if (k == MM_HASH_MAX) continue;
    // the cluster around the slot: MM_SK_GUARD neighbours on either side are fetched at once (independent LDS reads, one latency);
    // only a cluster that reaches further is walked slot by slot.  The guard slots beyond the table are always empty.
    uint64_t L[MM_SK_GUARD], Rr[MM_SK_GUARD];
#pragma unroll
    for (int i = 0; i < MM_SK_GUARD; i++) { L[i] = tab.key[slot - 1 - i]; Rr[i] = tab.key[slot + 1 + i]; }
    uint32_t smaller = 0;
    int q = slot;                                   // -> first slot of the cluster
    bool openL = true, openR = true;
#pragma unroll
    for (int i = 0; i < MM_SK_GUARD; i++) {
      openL = openL && L[i] != MM_HASH_MAX;
      openR = openR && Rr[i] != MM_HASH_MAX;
      if (openL) { smaller += L[i] < k ? 1u : 0u; q = slot - 1 - i; }
      if (openR) smaller += Rr[i] < k ? 1u : 0u;
    }
    if (openL) { while (q > 0) { const uint64_t o = tab.key[q - 1]; if (o == MM_HASH_MAX) break; smaller += o < k ? 1u : 0u; q--; } }
    if (openR) { for (int r = slot + 1 + MM_SK_GUARD; r < NS; r++) { const uint64_t o = tab.key[r]; if (o == MM_HASH_MAX) break; smaller += o < k ? 1u : 0u; } }
    const uint64_t below = tab.occW[q >> 6] & ((1ull << (q & 63)) - 1ull);
    const uint32_t rank = tab.occP[q >> 6] + (uint32_t)__popcll(below) + smaller;
    if (rank < (uint32_t)s) {
      const size_t o = (size_t)f * s + rank;
      skHash[o] = k;
      int vf, vl, vs;
      if constexpr (SPILL) {                          // written by device-scope atomics: read at the same scope, past the CU's L1
        vf = __hip_atomic_load(&tab.first[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vl = __hip_atomic_load(&tab.last[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vs = __hip_atomic_load(&tab.sum[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else { vf = tab.first[slot]; vl = tab.last[slot]; vs = tab.sum[slot]; }
      skPos[o] = make_int2(vf, vl);
      // the reference accumulates the strand in an int16 (base_types.hpp:24, commonFunc.hpp:268)
      const int16_t acc = (int16_t)vs;
      skStrand[o] = acc > 0 ? 1 : (acc == 0 ? 0 : -1);
    }
  }
  if (tid == 0) skCount[f] = D < (uint32_t)s ? D : (uint32_t)s;
}

template <int K, bool SPILL>
__global__ void __launch_bounds__(1024)
k_sketch_hard(const uint4* __restrict__ gTabs, const uint32_t* __restrict__ bases2, const uint32_t* __restrict__ nmask,
              const DFrag* __restrict__ frags, const uint32_t* __restrict__ readHasN,
              const int32_t* __restrict__ fragList, const uint32_t* __restrict__ fragListCount, int s, int wantFast, int HT, int PAD, int32_t* __restrict__ spill, int stream,
              uint64_t* __restrict__ skHash, int2* __restrict__ skPos, int8_t* __restrict__ skStrand, uint32_t* __restrict__ skCount) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the hard list's length stays on the device (no host round trip between the two kernels): a fixed grid walks it
  const uint32_t nList = *fragListCount;
  for (uint32_t i = blockIdx.x; i < nList; i += gridDim.x) {
    mm_sketch_hard<K, SPILL>(smem, fragList[i], gTabs, bases2, nmask, frags, readHasN, s, wantFast, HT, PAD, spill, stream != 0, skHash, skPos, skStrand, skCount);
    __syncthreads();                                // the next fragment reuses the LDS
  }
}

// every fragment on the hard list (sketches the fast kernel's LDS geometry cannot hold): no hint for the first cut
__global__ void __launch_bounds__(256)
k_sketch_all_hard(int nF, int32_t* __restrict__ hardList, uint32_t* __restrict__ hardCount, uint32_t* __restrict__ skCount) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f < nF) { hardList[f] = f; skCount[f] = MM_SK_HINT_OVERFLOW; }
  if (f == 0) *hardCount = (uint32_t)nF;
}

// ---------------------------------------------------------------------------------------------
// Fast sketch kernel, per fragment (one workgroup).  What paces it besides the hash loop is how many workgroups share a CU -- the
// table phases are LDS-latency bound and leave the VALU to whoever else is resident -- so the LDS footprint is kept small:
//   * a table slot is the 64-bit key plus ONE 32-bit word, the (position, strand) of the occurrence that claimed the slot.  Repeat
//     occurrences of a resident hash are rare outside repeats: they go on a short list, the slot gets a flag after the barrier, and
//     only flagged slots fold their list entries into first / last / strand sum when the sketch is written;
//   * the per-wave survivor queues hold the expected share of a wave plus six standard deviations, not a whole table's worth;
//   * SL positions per thread are chosen so that the threads fill whole waves (4 982 k-mers: 250 threads x 20 positions = 4 waves,
//     one per SIMD, instead of 312 x 16 = 5 waves of which two share a SIMD).
// A fragment that overflows any of it (queue, table, duplicate list) or keeps fewer than s distinct survivors goes to the hard list.
// ---------------------------------------------------------------------------------------------
#define MM_SKF_FLAG 0x80000000u
struct SkFast {
  uint64_t* key; uint32_t* meta;      // meta: bit 31 = has entries on the duplicate list, low bits = pos << 1 | (strand > 0) of the owner
  uint32_t* dup; uint32_t* counters;  // counters: [1] overflow  [2] occupied slots  [3] duplicates
  uint64_t* occW; uint32_t* occP;
  uint32_t nSlots, maxLoad, M; int sh;
  __device__ __forceinline__ void set_cut(uint64_t T, uint32_t HT) {
    sh = (T == MM_HASH_MAX) ? 0 : __clzll((long long)T);
    const uint64_t t32 = ((T << sh) >> 32) + 1ull;
    M = (uint32_t)((float)HT * 4294967296.0f / (float)t32 * 0.99999f);
  }
  __device__ __forceinline__ uint32_t home(uint64_t h) const { return __umulhi((uint32_t)((h << sh) >> 32), M); }
  __device__ __forceinline__ void insert(uint64_t h, uint32_t m) {
    uint32_t slot = home(h);
    for (;;) {
      const unsigned long long prev = atomicCAS((unsigned long long*)&key[slot], (unsigned long long)MM_HASH_MAX, (unsigned long long)h);
      if (prev == MM_HASH_MAX) { meta[slot] = m; return; }                     // the owner: a plain store, nobody reads it before the barrier
      if (prev == h) {
        const uint32_t at = atomicAdd(&counters[3], 1u);
        if (at < MM_SK_DUPCAP) dup[at] = slot | (m << 13); else atomicOr(&counters[1], 1u);
        return;
      }
      if (++slot >= nSlots) { atomicOr(&counters[1], 1u); return; }
    }
  }
};

template <int K, int SL>
__device__ __forceinline__ void
mm_sketch_fast(unsigned char* smem, const int f, const uint4* __restrict__ gTabs, const uint32_t* __restrict__ bases2, const uint32_t* __restrict__ nmask,
               const DFrag* __restrict__ frags, const uint32_t* __restrict__ readHasN, int s, int wantFast, int HT, int PAD, int QC,
               uint64_t* __restrict__ skHash, int2* __restrict__ skPos, int8_t* __restrict__ skStrand,
               uint32_t* __restrict__ skCount, int32_t* __restrict__ hardList, uint32_t* __restrict__ hardCount,
               unsigned long long* __restrict__ phaseStats) {
  // MM_SKETCH_STATS: shader-clock cycles thread 0 spends up to each phase boundary, summed over workgroups (diagnostics only)
  unsigned long long tPrev = phaseStats ? __builtin_amdgcn_s_memtime() : 0ull;
  auto mark = [&](int ph) {
    if (phaseStats && threadIdx.x == 0) { const unsigned long long t = __builtin_amdgcn_s_memtime(); atomicAdd(&phaseStats[ph], t - tPrev); tPrev = t; }
  };
  const DFrag fr = frags[f];
  const int len = fr.len;
  const int n = len - K + 1;                        // k-mer positions
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (n <= 0) { if (tid == 0) skCount[f] = 0; return; }
  const bool hasN = readHasN[fr.readId] != 0;

  // ---- LDS carve (every offset a multiple of 16) ----
  const int nW = (len + 15) / 16 + 4;               // code words incl. run-off for the last strip's window
  const int nM = (len + 31) / 32 + 3;
  const int NS = HT + PAD;                          // table slots (multiple of 64)
  const int nOcc = NS >> 6;
  const int nWaves = nthr >> 6;
  const int QTOT = sketch_fast_queue_total(QC, nWaves, HT);   // >= the table's load limit: the memory serves as the list of occupied slots later
  size_t off = 0;
  using Tabs = typename MMTabsFor<K>::type;
  Tabs* tabs = (Tabs*)(smem + off); off += sizeof(Tabs);
  for (int i = tid; i < (int)(sizeof(Tabs) / 16); i += nthr) ((uint4*)tabs)[i] = gTabs[i];   // visible after the first __syncthreads() below
  uint32_t* sW = (uint32_t*)(smem + off); off += (((size_t)nW * 4 + 15) / 16) * 16;
  uint32_t* sM = (uint32_t*)(smem + off); off += (((size_t)nM * 4 + 15) / 16) * 16;
  uint64_t* qH = (uint64_t*)(smem + off); off += (size_t)QC * nWaves * 8;        // per-wave queues: hashes
  uint32_t* qM = (uint32_t*)(smem + off); off += (size_t)QTOT * 4;               //                  pos<<1 | (strand > 0)
  SkFast tab;
  tab.key = (uint64_t*)(smem + off) + MM_SK_GUARD; off += (size_t)(NS + 2 * MM_SK_GUARD) * 8;
  tab.occW = (uint64_t*)(smem + off); off += (size_t)nOcc * 8;
  tab.meta = (uint32_t*)(smem + off); off += (size_t)NS * 4;
  tab.occP = (uint32_t*)(smem + off); off += (((size_t)nOcc * 4 + 15) / 16) * 16;
  uint32_t* qCnt = (uint32_t*)(smem + off); off += 64;
  tab.dup = (uint32_t*)(smem + off); off += (size_t)MM_SK_DUPCAP * 4;
  tab.counters = (uint32_t*)(smem + off); off += 16;
  tab.nSlots = (uint32_t)NS; tab.maxLoad = (uint32_t)HT * 5u / 8u;

  // ---- stage the fragment, re-aligned so that LDS word j holds bases 16j..16j+15 ----
  {
    const int64_t w0 = fr.base >> 4; const int sh = (int)(fr.base & 15) * 2;
    for (int j = tid; j < nW; j += nthr) {
      const uint32_t a = bases2[w0 + j], b = bases2[w0 + j + 1];
      sW[j] = sh ? __builtin_amdgcn_alignbit(b, a, sh) : a;
    }
    if (hasN) {
      const int64_t m0 = fr.base >> 5; const int msh = (int)(fr.base & 31);
      for (int j = tid; j < nM; j += nthr) {
        const uint32_t a = nmask[m0 + j], b = nmask[m0 + j + 1];
        sM[j] = msh ? __builtin_amdgcn_alignbit(b, a, msh) : a;
      }
    }
  }
  // threshold: the canonical hash is the smaller of two uniform 64-bit values, so P[h < T] ~ 2T / 2^64; the cut is placed where
  // `wantFast` (> s) survivors are expected.  Any T gives the exact sketch as long as >= s distinct hashes survive (checked below).
  const uint64_t T = ((uint32_t)wantFast >= (uint32_t)n) ? MM_HASH_MAX : (uint64_t)((float)wantFast / (float)(2 * n) * 18446744073709551616.0f);
  const int nStrips = (n + SL - 1) / SL;
  tab.set_cut(T, (uint32_t)HT);
  for (int i = tid; i < NS; i += nthr) tab.key[i] = MM_HASH_MAX;
  if (tid < MM_SK_GUARD) { tab.key[-1 - tid] = MM_HASH_MAX; tab.key[NS + tid] = MM_HASH_MAX; }
  if (tid < 4) tab.counters[tid] = 0;
  __syncthreads();
  mark(0);                                          // staging + table init

  // ---- phase 1: hash both strands of every k-mer ----
  const uint32_t qBase = (uint32_t)(tid >> 6) * (uint32_t)QC, qEnd = qBase + (uint32_t)QC;
  uint32_t qHead = qBase;                           // wave-uniform (only ballots feed it)
  const uint64_t allPassM = (T == MM_HASH_MAX) ? ~0ull : 0ull;     // wave-uniform
  for (int strip = tid; strip < nStrips; strip += nthr) {
    const int b0 = strip * SL;                      // first base of the strip: the 48-base window is cut out of four LDS words
    const int wi = b0 >> 4, bsh = (b0 & 15) * 2;
    // bit j: position b0 + j exists and its k-mer holds no N
    const int rem = n - b0;
    uint32_t ok = rem >= SL ? (uint32_t)((1ull << SL) - 1ull) : ((1u << rem) - 1u);
    if (hasN) {
      const int mi = b0 >> 5, msh = b0 & 31;
      const uint64_t lo64 = (uint64_t)sM[mi] | ((uint64_t)sM[mi + 1] << 32);
      if constexpr (MMWideK<K>::value) {            // 16 + K - 1 > 64 mask bits: a 128-bit window
        const uint64_t hi64 = (uint64_t)sM[mi + 2] | ((uint64_t)sM[mi + 3] << 32);
        const uint64_t lo = msh ? ((lo64 >> msh) | (hi64 << (64 - msh))) : lo64, hi = msh ? (hi64 >> msh) : hi64;
        ok &= ~mm_window_or_wide<K>(lo, hi);
      } else {
        const uint64_t m64 = msh ? ((lo64 >> msh) | ((uint64_t)sM[mi + 2] << (64 - msh))) : lo64;
        ok &= ~(uint32_t)mm_window_or<K>(m64);
      }
    }
    auto onPos = [&](int j, uint64_t hf, uint64_t hr) {
      const int pos = b0 + j;
      // the tests as lane masks in scalar registers (mm_device.h): nothing but the two selects of the minimum touches a vector register
      // (the compiler's own `hf < hr ? hf : hr` is as fast at s = 130 and 0.5 ms per 2 M fragments slower at s = 310: profiles/r06b, r06c)
      const uint64_t mLt = mm_mask_lt64(hf, hr);
      const uint64_t h = mm_mask_select64(mLt, hf, hr);
      const uint64_t m = mm_mask_nz32(ok & (1u << j)) & mm_mask_ne64(hf, hr) & (mm_mask_lt64(h, T) | allPassM);
      const uint32_t idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, qHead));
      if (__builtin_amdgcn_inverse_ballot_w64(m) && idx < qEnd) { qH[idx] = h; qM[idx] = ((uint32_t)pos << 1) | (__builtin_amdgcn_inverse_ballot_w64(mLt) ? 1u : 0u); }
      qHead += (uint32_t)__popcll(m);
    };
    if constexpr (MMWideK<K>::value) {              // k-mers of 33..64 bases: 4 or 5 words per strip (SL == 16: the strip starts on a word)
      uint32_t ww[MMWideK<K>::NW];
#pragma unroll
      for (int i = 0; i < MMWideK<K>::NW; i++) ww[i] = sW[wi + i];
      mm_strip_hashes_wide<K>(ww, *tabs, onPos);
    } else {
      uint32_t w0 = sW[wi], w1 = sW[wi + 1], w2 = sW[wi + 2];
      if (SL % 16 != 0) {
        const uint32_t w3 = sW[wi + 3];
        if (bsh) { w0 = __builtin_amdgcn_alignbit(w1, w0, bsh); w1 = __builtin_amdgcn_alignbit(w2, w1, bsh); w2 = __builtin_amdgcn_alignbit(w3, w2, bsh); }
      }
      mm_strip_hashes<K, SL>(w0, w1, w2, *tabs, onPos);
    }
  }
  mark(1);                                          // thread 0's hash loop
  {
    // lanes that left the strip loop early (or never entered it) hold a stale count; lane 0 owns the smallest strip
    const uint32_t qCount = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qHead - qBase));
    if (mm_lane() == 0) { qCnt[tid >> 6] = qCount; if (qCount > (uint32_t)QC) atomicOr(&tab.counters[1], 1u); }
  }
  __syncthreads();
  mark(2);                                          // waiting for the other waves' hash loops
  if (tab.counters[1] == 0) {
    // all threads drain all queues: entry g of the concatenated queues goes to thread g mod nthr, so every round of inserts is full
    uint32_t pre[17]; pre[0] = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) pre[w + 1] = pre[w] + (w < nWaves ? qCnt[w] : 0u);
    const uint32_t total = pre[16];
    if (nWaves <= 4) {
      // the usual geometry (4 982 k-mers = 250 threads = 4 waves): three compares place an entry instead of the fifteen of the general form
      // below, which were a sixth of the instructions a survivor costs behind the hash loop
      const uint32_t p1 = pre[1], p2 = pre[2], p3 = pre[3];
      for (uint32_t g = (uint32_t)tid; g < total; g += (uint32_t)nthr) {
        const uint32_t w = (g >= p1 ? 1u : 0u) + (g >= p2 ? 1u : 0u) + (g >= p3 ? 1u : 0u);
        const uint32_t start = g >= p3 ? p3 : (g >= p2 ? p2 : (g >= p1 ? p1 : 0u));
        const uint32_t at = w * (uint32_t)QC + (g - start);
        tab.insert(qH[at], qM[at]);
      }
    } else
    for (uint32_t g = (uint32_t)tid; g < total; g += (uint32_t)nthr) {
      uint32_t w = 0, start = 0;                    // the queue entry g falls into, and where that queue starts in the concatenation
#pragma unroll
      for (int x = 1; x < 16; x++) if (g >= pre[x]) { w = (uint32_t)x; start = pre[x]; }
      const uint32_t at = w * (uint32_t)QC + (g - start);
      tab.insert(qH[at], qM[at]);
    }
  }
  __syncthreads();
  mark(3);                                          // queues drained into the table
  {
    // owners' stores are visible now: flag the slots that have entries on the duplicate list
    const uint32_t nd = tab.counters[3] < MM_SK_DUPCAP ? tab.counters[3] : MM_SK_DUPCAP;
    for (uint32_t i = (uint32_t)tid; i < nd; i += (uint32_t)nthr) atomicOr(&tab.meta[tab.dup[i] & 0x1FFFu], MM_SKF_FLAG);
  }
  // ---- occupancy words, their prefix counts (wave 0: one DPP scan per 64 words), number of distinct keys ----
  for (int base = 0; base < NS; base += nthr) {
    const int slot = base + tid;                    // NS and nthr are multiples of 64: a wave is in or out as a whole
    if (slot < NS) {
      const uint64_t m = __ballot(tab.key[slot] != MM_HASH_MAX);
      if (mm_lane() == 0) tab.occW[slot >> 6] = m;
    }
  }
  __syncthreads();
  if (tid < 64) {
    int carry = 0;
    for (int w0 = 0; w0 < nOcc; w0 += 64) {
      const int w = w0 + tid;
      const int v = w < nOcc ? (int)__popcll(tab.occW[w]) : 0;
      const int ex = mm_wave_excl_scan(v);
      if (w < nOcc) tab.occP[w] = (uint32_t)(carry + ex);
      carry += mm_wave_sum(v);
    }
    if (tid == 0) { tab.counters[2] = (uint32_t)carry; if ((uint32_t)carry > tab.maxLoad) tab.counters[1] = 1u; }
  }
  __syncthreads();
  const uint32_t D = tab.counters[2];
  if (tab.counters[1] != 0 || (D < (uint32_t)s && T != MM_HASH_MAX)) {
    // skCount[f] carries the first hint for the hard kernel's cut: the distinct count under this one, unless something overflowed
    if (tid == 0) { const uint32_t at = atomicAdd(hardCount, 1u); hardList[at] = f; skCount[f] = tab.counters[1] != 0 ? MM_SK_HINT_OVERFLOW : D; }
    return;
  }
  // the occupied slots in slot order (the queue memory is free again): every thread then ranks D / nthr keys instead of visiting
  // NS / nthr slots of which two thirds are empty; a key's position in that list is the number of occupied slots before it
  uint32_t* occList = qM;
  for (int slot = tid; slot < NS; slot += nthr) {
    const uint64_t m = tab.occW[slot >> 6];
    if ((m >> (slot & 63)) & 1ull) occList[tab.occP[slot >> 6] + (uint32_t)__popcll(m & ((1ull << (slot & 63)) - 1ull))] = (uint32_t)slot;
  }
  __syncthreads();
  mark(4);                                          // duplicate flags, occupancy, prefix counts, list of occupied slots

  // ---- rank every key inside its cluster; emit the s smallest, ascending (commonFunc.hpp:278-286) ----
  const uint32_t nDup = tab.counters[3];
  for (int it = tid; it < (int)D; it += nthr) {
    const int slot = (int)occList[it];
    const uint64_t k = tab.key[slot];
    // the cluster around the slot: MM_SK_GUARD neighbours on either side are fetched at once (independent LDS reads, one latency);
    // only a cluster that reaches further is walked slot by slot.  The guard slots beyond the table are always empty.
    uint64_t L[MM_SK_GUARD], Rr[MM_SK_GUARD];
#pragma unroll
    for (int i = 0; i < MM_SK_GUARD; i++) { L[i] = tab.key[slot - 1 - i]; Rr[i] = tab.key[slot + 1 + i]; }
    uint32_t smaller = 0;
    int q = slot;                                   // -> first slot of the cluster
    bool openL = true, openR = true;
#pragma unroll
    for (int i = 0; i < MM_SK_GUARD; i++) {
      openL = openL && L[i] != MM_HASH_MAX;
      openR = openR && Rr[i] != MM_HASH_MAX;
      if (openL) { smaller += L[i] < k ? 1u : 0u; q = slot - 1 - i; }
      if (openR) smaller += Rr[i] < k ? 1u : 0u;
    }
    if (openL) { while (q > 0) { const uint64_t o = tab.key[q - 1]; if (o == MM_HASH_MAX) break; smaller += o < k ? 1u : 0u; q--; } }
    if (openR) { for (int r = slot + 1 + MM_SK_GUARD; r < NS; r++) { const uint64_t o = tab.key[r]; if (o == MM_HASH_MAX) break; smaller += o < k ? 1u : 0u; } }
    const uint32_t rank = (uint32_t)(it - (slot - q)) + smaller;               // every slot of [q, slot) is occupied
    if (rank < (uint32_t)s) {
      const uint32_t m = tab.meta[slot];
      int first = (int)((m & ~MM_SKF_FLAG) >> 1), last = first, sum = (m & 1u) ? 1 : -1;
      if (m & MM_SKF_FLAG) {                                                   // repeat occurrences: first / last position, strand sum (commonFunc.hpp:242-270)
        for (uint32_t i = 0; i < nDup; i++) {
          const uint32_t d = tab.dup[i];
          if ((int)(d & 0x1FFFu) != slot) continue;
          const int pos = (int)(d >> 14);
          first = pos < first ? pos : first; last = pos > last ? pos : last; sum += ((d >> 13) & 1u) ? 1 : -1;
        }
      }
      const size_t o = (size_t)f * s + rank;
      skHash[o] = k;
      skPos[o] = make_int2(first, last);
      // the reference accumulates the strand in an int16 (base_types.hpp:24, commonFunc.hpp:268)
      const int16_t acc = (int16_t)sum;
      skStrand[o] = acc > 0 ? 1 : (acc == 0 ? 0 : -1);
    }
  }
  if (tid == 0) skCount[f] = D < (uint32_t)s ? D : (uint32_t)s;
  mark(5);                                          // ranking + output (thread 0's share)
}

template <int K, int SL>
__global__ void __launch_bounds__(1024)
k_sketch_fast(const uint4* __restrict__ gTabs, const uint32_t* __restrict__ bases2, const uint32_t* __restrict__ nmask,
              const DFrag* __restrict__ frags, const uint32_t* __restrict__ readHasN, int s, int wantFast, int HT, int PAD, int QC,
              uint64_t* __restrict__ skHash, int2* __restrict__ skPos, int8_t* __restrict__ skStrand,
              uint32_t* __restrict__ skCount, int32_t* __restrict__ hardList, uint32_t* __restrict__ hardCount,
              unsigned long long* __restrict__ phaseStats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  mm_sketch_fast<K, SL>(smem, (int)blockIdx.x, gTabs, bases2, nmask, frags, readHasN, s, wantFast, HT, PAD, QC, skHash, skPos, skStrand, skCount,
                        hardList, hardCount, phaseStats);
}

// ---------------------------------------------------------------------------------------------
// k_hash_only<K, SL>: the integer roofline's yardstick, at the fast sketch kernel's OWN geometry: workgroup per
// fragment, the same thread count, the same SL positions per thread (window cut out of four LDS words), the same LDS claim (so the same
// number of workgroups share a CU), packed words and hasher tables in LDS -- but nothing besides the 2 x MurmurHash3_x64_128 per
// position: no N mask, no cut, no queues, no table.  The canonical hashes are folded into one value per thread that is stored only if
// it equals an impossible constant, so the hashes stay live and nothing is written.
// ---------------------------------------------------------------------------------------------
template <int K, int SL>
__global__ void __launch_bounds__(1024)
k_hash_only(const uint4* __restrict__ gTabs, const uint32_t* __restrict__ bases2, const DFrag* __restrict__ frags, uint64_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const DFrag fr = frags[blockIdx.x];
  const int len = fr.len, n = len - K + 1;
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (n <= 0) return;
  using Tabs = typename MMTabsFor<K>::type;
  Tabs* tabs = (Tabs*)smem;
  for (int i = tid; i < (int)(sizeof(Tabs) / 16); i += nthr) ((uint4*)tabs)[i] = gTabs[i];
  uint32_t* sW = (uint32_t*)(smem + sizeof(Tabs));
  const int nW = (len + 15) / 16 + 4;
  {
    const int64_t w0 = fr.base >> 4; const int sh = (int)(fr.base & 15) * 2;
    for (int j = tid; j < nW; j += nthr) {
      const uint32_t a = bases2[w0 + j], b = bases2[w0 + j + 1];
      sW[j] = sh ? __builtin_amdgcn_alignbit(b, a, sh) : a;
    }
  }
  __syncthreads();
  const int nStrips = (n + SL - 1) / SL;
  uint64_t acc = 0;
  for (int strip = tid; strip < nStrips; strip += nthr) {
    const int b0 = strip * SL;
    const int wi = b0 >> 4, bsh = (b0 & 15) * 2;
    uint32_t w0 = sW[wi], w1 = sW[wi + 1], w2 = sW[wi + 2];
    if (SL % 16 != 0) {
      const uint32_t w3 = sW[wi + 3];
      if (bsh) { w0 = __builtin_amdgcn_alignbit(w1, w0, bsh); w1 = __builtin_amdgcn_alignbit(w2, w1, bsh); w2 = __builtin_amdgcn_alignbit(w3, w2, bsh); }
    }
    mm_strip_hashes<K, SL>(w0, w1, w2, *tabs, [&](int j, uint64_t hf, uint64_t hr) { acc += hf < hr ? hf : hr; });
  }
  if (acc == 0x9E3779B97F4A7C15ull) sink[0] = acc;
}

// strip-hasher tables, built once per context and k-mer size
template <int K>
__global__ void k_sketch_tables(typename MMTabsFor<K>::type* out) { mm_tables_init<K>(*out, (int)threadIdx.x, (int)blockDim.x); }

// ---------------------------------------------------------------------------------------------
static size_t sketch_hard_lds(size_t tabBytes, int maxLen, int HT, int PAD) {      // the carve of mm_sketch_hard
  const size_t nW = (size_t)(maxLen + 15) / 16 + 3, nM = (size_t)(maxLen + 31) / 32 + 2;
  const size_t NS = (size_t)HT + PAD, nOcc = NS / 64;
  return tabBytes + ((nW * 4 + 15) / 16) * 16 + ((nM * 4 + 15) / 16) * 16 + NS * (8 + 4 + 4 + 4) + 2 * MM_SK_GUARD * 8 + nOcc * 8 +
         ((nOcc * 4 + 15) / 16) * 16 + 16;
}
static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// geometry of the fast kernel for fragments of up to maxLen bases: positions per thread, threads, table and queue sizes
struct FastGeom { int SL, threads, HT, QC, wantFast; size_t lds; };
static size_t sketch_fast_lds(size_t tabBytes, int maxLen, int HT, int PAD, int QC, int nWaves) {
  const size_t nW = (size_t)(maxLen + 15) / 16 + 4, nM = (size_t)(maxLen + 31) / 32 + 3;
  const size_t NS = (size_t)HT + PAD, nOcc = NS / 64;
  return tabBytes + ((nW * 4 + 15) / 16) * 16 + ((nM * 4 + 15) / 16) * 16 + (size_t)QC * nWaves * 8 + (size_t)sketch_fast_queue_total(QC, nWaves, HT) * 4 +
         (NS + 2 * MM_SK_GUARD) * 8 + nOcc * 8 + NS * 4 + ((nOcc * 4 + 15) / 16) * 16 + 64 + (size_t)MM_SK_DUPCAP * 4 + 16;
}
static FastGeom sketch_fast_geom(int K, int s, int maxLen, size_t tabBytes, bool sl20Built) {
  FastGeom g;
  // survivors the fast kernel aims for: s + 4.2 sqrt(s) (their count is ~Poisson, so s stays about 4 sigma away): fewer = less queue /
  // table work and less LDS, more fragments redone by k_sketch_hard (measured, profiles/r02v_sketch_geometry.txt: 0.007 % of random
  // fragments at s = 130, 0.008 % at 310, 0.02 % at 498, at 44 ns each).  MM_SKETCH_CUT = wanted survivors / s.
  g.wantFast = (int)(s + 4.2 * sqrt((double)s) + 0.999);
  if (const char* e = getenv("MM_SKETCH_CUT")) { double cut = atof(e); if (cut < 1.05) cut = 1.05; if (cut > 2.5) cut = 2.5; g.wantFast = (int)(s * cut + 0.999); }
  int n = maxLen - K + 1; if (n < 1) n = 1;
  // positions per thread: 16, or 20 where that fills whole waves better (the workgroup's critical path is ceil(waves / 4 SIMDs) strips)
  auto wavesFor = [&](int SL) { int st = (n + SL - 1) / SL; int w = (st + 63) / 64; return w < 1 ? 1 : (w > 16 ? 16 : w); };
  g.SL = 16;
  if (sl20Built && 20 + K - 1 <= 48) {
    const int w16 = wavesFor(16), w20 = wavesFor(20);
    const int c16 = ((w16 + 3) / 4) * 16, c20 = ((w20 + 3) / 4) * 20;
    if (c20 < c16 || (c20 == c16 && w20 < w16)) g.SL = 20;
  }
  if (const char* e = getenv("MM_SKETCH_SL")) { const int v = atoi(e); if (v == 16 || (v == 20 && sl20Built && 20 + K - 1 <= 48)) g.SL = v; }
  g.threads = wavesFor(g.SL) * 64;
  if (const char* e = getenv("MM_SKETCH_THREADS")) { const int t = atoi(e); if (t >= 64 && t <= 1024 && t % 64 == 0) g.threads = t; }
  const int nWaves = g.threads / 64;
  // a wave queues the survivors of its 64 threads' strips: their expectation + qsig standard deviations; the table has htf slots per
  // wanted survivor (load limit = 5/8 of them).  The kernel's table phases wait on LDS latency and are hidden by the other workgroups
  // of the CU, whose number the LDS footprint decides (allocated in 1280-byte units out of 160 KB): the roomy geometry is kept unless a
  // tighter one lets one more workgroup in (s = 130: 7 instead of 6, s = 310: 5 instead of 4, s = 498: 4 instead of 3 -- 49 -> 42 ms per
  // 1.6 M fragments there).  MM_SKETCH_HTF / MM_SKETCH_QSIG pin the two numbers.
  int nStrips = (n + g.SL - 1) / g.SL; if (nStrips < 1) nStrips = 1;
  const int passes = (nStrips + g.threads - 1) / g.threads;
  double ex = 64.0 * passes * g.SL * (double)g.wantFast / (double)n; if (ex > g.wantFast) ex = g.wantFast;
  auto shape = [&](double htf, double qsig) {
    int HT = (int)(htf * g.wantFast + 63) / 64 * 64; if (HT < 256) HT = 256;
    g.HT = HT;
    g.QC = ((int)(ex + qsig * sqrt(ex) + 8.0) + 3) & ~3;
    g.lds = sketch_fast_lds(tabBytes, maxLen, g.HT, MM_SK_PAD, g.QC, nWaves);
    const size_t unit = 1280, alloc = (g.lds + unit - 1) / unit * unit;
    return (int)((size_t)160 * 1024 / alloc);      // workgroups a CU holds
  };
  const char* eh = getenv("MM_SKETCH_HTF"); const char* eq = getenv("MM_SKETCH_QSIG");
  if (eh || eq) {
    double htf = eh ? atof(eh) : 2.4, qsig = eq ? atof(eq) : 6.0;
    if (htf < 1.8) htf = 1.8; if (htf > 4.0) htf = 4.0; if (qsig < 3.0) qsig = 3.0; if (qsig > 8.0) qsig = 8.0;
    (void)shape(htf, qsig);
  } else {
    const double cand[3][2] = {{2.4, 6.0}, {2.2, 6.0}, {2.2, 5.0}};
    int best = 0, bestWg = shape(cand[0][0], cand[0][1]);
    for (int i = 1; i < 3; i++) { const int wg = shape(cand[i][0], cand[i][1]); if (wg > bestWg) { bestWg = wg; best = i; } }
    (void)shape(cand[best][0], cand[best][1]);
  }
  return g;
}
// hard kernel's table: the load limit, 5/8 of it, stays >= 2 s (the window [s, limit] the cut must hit is an octave wide); no larger
// than that needs, because its LDS sets how many listed fragments a CU works on at once (s = 130: 4 workgroups)
static int sketch_ht_hard(int s) { const int w = (s * 16 + 4) / 5; return next_pow2(w < 1024 ? 1024 : w); }

// How the two sketch kernels are run for (k, s, fragment length): the fast kernel when its geometry fits (LDS, and table slots that fit
// the 13-bit field of its duplicate list), otherwise every fragment goes down the hard list; the hard kernel with its whole table in
// LDS when that fits, otherwise with the first / last / strand-sum arrays spilled to HBM scratch and, if the power-of-two table still
// does not fit, the smallest table that keeps its load limit (5/8) at 2 s.  ok == false: not even that fits (the message says why).
struct SketchPlan { bool ok, useFast, spill, stream = false; FastGeom g; int HTH; size_t ldsFast, ldsHard; };
static size_t sketch_hard_lds_spill(size_t tabBytes, int maxLen, int HT, int PAD) {
  const size_t nW = (size_t)(maxLen + 15) / 16 + 3, nM = (size_t)(maxLen + 31) / 32 + 2;
  const size_t NS = (size_t)HT + PAD, nOcc = NS / 64;
  return tabBytes + ((nW * 4 + 15) / 16) * 16 + ((nM * 4 + 15) / 16) * 16 + NS * 8 + 2 * MM_SK_GUARD * 8 + nOcc * 8 + ((nOcc * 4 + 15) / 16) * 16 + 16;
}
static SketchPlan sketch_plan(int K, int s, int maxLen, size_t tabBytes, bool sl20Built) {
  const size_t lim = 160 * 1024;
  SketchPlan P;
  P.g = sketch_fast_geom(K, s, maxLen, tabBytes, sl20Built);
  P.ldsFast = P.g.lds;
  P.useFast = P.ldsFast <= lim && P.g.HT + MM_SK_PAD <= 8192 && maxLen < (1 << 18) && !getenv("MM_SKETCH_ALL_HARD");
  P.HTH = sketch_ht_hard(s);
  P.spill = getenv("MM_SKETCH_SPILL") != nullptr;
  P.ldsHard = sketch_hard_lds(tabBytes, maxLen, P.HTH, MM_SK_PADH);
  if (P.ldsHard > lim || P.spill) {
    P.spill = true;
    P.ldsHard = sketch_hard_lds_spill(tabBytes, maxLen, P.HTH, MM_SK_PADH);
    if (P.ldsHard > lim) {                          // not a power of two then: load limit 5/8 HT >= 2 s
      P.HTH = (((s * 16 + 4) / 5) + 63) / 64 * 64;
      P.ldsHard = sketch_hard_lds_spill(tabBytes, maxLen, P.HTH, MM_SK_PADH);
    }
  }
  if (P.ldsHard > lim) {                            // the staged fragment is what does not fit (a whole contig as one fragment under --noSplit):
    P.stream = true;                                // the exact kernel reads its words from global memory instead
    P.HTH = sketch_ht_hard(s);
    P.ldsHard = sketch_hard_lds_spill(tabBytes, 0, P.HTH, MM_SK_PADH);
    if (P.ldsHard > lim) { P.HTH = (((s * 16 + 4) / 5) + 63) / 64 * 64; P.ldsHard = sketch_hard_lds_spill(tabBytes, 0, P.HTH, MM_SK_PADH); }
  }
  P.ok = P.ldsHard <= lim;
  return P;
}

// Parameter combinations the LDS-resident kernels cannot hold are refused when the context is created (not after the reference
// index has been built): the sketch tables + a staged fragment of segLength bases, and the 16-bit L2 state cells of k_l2_sweep.
int mm_check_params(const mm_params* p, std::string& err) {
  const int s = p->sketchSize, L = p->segLength;
  const size_t tabBytes = (p->kmerSize >= 16 && p->kmerSize <= 32) ? sizeof(MMProdTables) : sizeof(MMTables);
  const SketchPlan P = sketch_plan(p->kmerSize, s, L, tabBytes, false);
  const size_t lim = 160 * 1024;
  // L2: the 16-bit state cells of as few as 8 candidates per wave, the query sketch + bucket table of one wave of k_l2_locate, and the
  // 13-bit sketch-position field of the located stream (mm_l2.hip)
  const size_t ldsL2 = (size_t)(s + 1) * 8 * 2;
  int NB = 256; while (NB < s) NB <<= 1;
  const size_t ldsLoc = (size_t)(s + 1) * 8 + (size_t)(s + 2) / 2 * 8 + (size_t)(NB + 4) * 2 + (size_t)s + 32 + (size_t)(s + 4) / 2 * 4 + 16;   // (one wave of k_l2_locate: mm_locate_lds_per_wave)
  if (s > MM_LDS_MAX_SKETCH) {
    // no LDS kernel holds this sketch: the global-memory sketch kernel (mm_sketch_global.hip) and the literal L2 kernels take every
    // fragment -- exact, slow; the stock binary runs these sizes (--dense at segments of 100 kbp), so they run here too
    // ... and so does the index build: k_winnow_tiles (mm_winnow.hip) keeps the sketch of a reference window in LDS up to
    // MM_WINNOW_LDS_SKETCH entries and in HBM beyond
    if (s > MM_MAX_SKETCH) {
      err = "mm_create: sketchSize " + std::to_string(s) + " is beyond " + std::to_string(MM_MAX_SKETCH) + " (seeds are numbered in 16 bits by the literal mapping kernels)";
      return MM_ERR_ARG;
    }
    return MM_OK;
  }
  if (!P.ok || ldsL2 > lim || ldsLoc > lim) {
    char b[400];
    snprintf(b, sizeof b, "mm_create: segLength %d with sketchSize %d needs %zu bytes of LDS in the sketch kernel (table of the exact path, spilled form), %zu in the L2 "
             "sweep and %zu in the L2 locate kernel; a CU has %zu (sketchSize up to ~5000)",
             L, s, P.ldsHard, ldsL2, ldsLoc, lim);
    err = b; return MM_ERR_ARG;
  }
  return MM_OK;
}

template <int K> struct MMHasSL20 { static constexpr bool value = (K >= 16 && K <= 21); };   // strips of 20 positions are built for these k-mer sizes

template <int K>
static int launch_sketch_k(mm_ctx* c) {
  const int s = c->P.sketchSize;
  const int nF = (int)c->nFrags;
  using Tabs = typename MMTabsFor<K>::type;
  const int maxLen = c->maxFragLen;
  const SketchPlan plan = sketch_plan(K, s, maxLen, sizeof(Tabs), MMHasSL20<K>::value);
  const FastGeom g = plan.g;
  const int HTH = plan.HTH;
  const int PAD = MM_SK_PAD, PADH = MM_SK_PADH;     // spill slots behind the ordered tables (no wrap-around)
  const size_t ldsFast = plan.ldsFast, ldsHard = plan.ldsHard;
  if (!plan.ok) { c->err = "fragment too long / sketch too large for the LDS-resident sketch kernel"; return MM_ERR_ARG; }
  int nStrips = (maxLen - K + 1 + 15) / 16; if (nStrips < 1) nStrips = 1;
  int threadsHard = ((nStrips + 63) / 64) * 64; if (threadsHard > 1024) threadsHard = 1024;
  if (c->sketchTabsK != K) {
    MM_HIP(c, c->dSketchTabs.ensure(sizeof(Tabs)));
    hipLaunchKernelGGL((k_sketch_tables<K>), dim3(1), dim3(256), 0, c->stream, c->dSketchTabs.as<Tabs>());
    MM_HIP(c, hipGetLastError());
    c->sketchTabsK = K;
  }
  MM_HIP(c, hipMemsetAsync(c->dCounters.p, 0, 64, c->stream));
  unsigned long long* phaseStats = nullptr;
  if (getenv("MM_SKETCH_STATS")) { phaseStats = c->dCounters.as<unsigned long long>() + 24; MM_HIP(c, hipMemsetAsync(phaseStats, 0, 64, c->stream)); }
  // MM_SKETCH_PROBE=1: the fast kernel probes the seed table for the sketch it emits (needs the queue memory to hold s values)
  if (plan.useFast) {
    KernelTimer t(c, MM_K_SKETCH);
    auto launch = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsFast);
      hipLaunchKernelGGL(kern, dim3(nF), dim3(g.threads), ldsFast, c->stream,
                         c->dSketchTabs.as<uint4>(), c->dBases2.as<uint32_t>(), c->dNmask.as<uint32_t>(), c->dFrags.as<DFrag>(), c->dReadHasN.as<uint32_t>(),
                         s, g.wantFast, g.HT, PAD, g.QC, c->dSkHash.as<uint64_t>(), c->dSkPos.as<int2>(), c->dSkStrand.as<int8_t>(),
                         c->dSkCount.as<uint32_t>(), c->dHardList.as<int32_t>(), c->dCounters.as<uint32_t>(), phaseStats);
    };
    if constexpr (MMHasSL20<K>::value) { if (g.SL == 20) launch(k_sketch_fast<K, 20>); else launch(k_sketch_fast<K, 16>); }
    else launch(k_sketch_fast<K, 16>);
    MM_HIP(c, hipGetLastError());
  } else {
    // the fast kernel's LDS geometry cannot hold this sketch: every fragment takes the exact path
    KernelTimer t(c, MM_K_SKETCH);
    hipLaunchKernelGGL(k_sketch_all_hard, dim3((nF + 255) / 256), dim3(256), 0, c->stream, nF, c->dHardList.as<int32_t>(), c->dCounters.as<uint32_t>(), c->dSkCount.as<uint32_t>());
    MM_HIP(c, hipGetLastError());
  }
  if (phaseStats && plan.useFast) {
    unsigned long long h[8];
    MM_HIP(c, hipMemcpyAsync(h, phaseStats, 64, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
    fprintf(stderr, "[mm] sketch phases, shader-clock cycles per workgroup (thread 0), %d fragments, %d threads x %d positions, HT %d, queue %d/wave, lds %zu: stage+init %.0f | hash %.0f | wait %.0f | drain %.0f | occupancy+list %.0f | rank+emit %.0f\n",
            nF, g.threads, g.SL, g.HT, g.QC, ldsFast, (double)h[0] / nF, (double)h[1] / nF, (double)h[2] / nF, (double)h[3] / nF, (double)h[4] / nF, (double)h[5] / nF);
  }
  if (getenv("MM_DEBUG")) {
    uint32_t nHard = 0;
    MM_HIP(c, hipMemcpyAsync(&nHard, c->dCounters.p, 4, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
    fprintf(stderr, "[mm] sketch: %d fragments, %u to the hard path%s, threads %d x %d positions, HT %d, lds %zu/%zu%s\n", nF, nHard, plan.useFast ? "" : " (fast kernel not usable at this size)",
            g.threads, g.SL, g.HT, ldsFast, ldsHard, plan.stream ? " (hard table: position / strand arrays in HBM; fragments read from global memory)" : plan.spill ? " (hard table: position / strand arrays in HBM)" : "");
  }
  {
    // fixed grid over the device-resident hard list (its workgroups leave at once when the list is empty or short)
    KernelTimer t(c, MM_K_SKETCH_HARD);
    const int grid = nF < 1024 ? nF : 1024;
    auto launchHard = [&](auto kern, int32_t* spill) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsHard);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(threadsHard), ldsHard, c->stream,
                         c->dSketchTabs.as<uint4>(), c->dBases2.as<uint32_t>(), c->dNmask.as<uint32_t>(), c->dFrags.as<DFrag>(), c->dReadHasN.as<uint32_t>(),
                         c->dHardList.as<int32_t>(), c->dCounters.as<uint32_t>(), s, g.wantFast, HTH, PADH, spill, plan.stream ? 1 : 0, c->dSkHash.as<uint64_t>(), c->dSkPos.as<int2>(), c->dSkStrand.as<int8_t>(),
                         c->dSkCount.as<uint32_t>());
    };
    if (plan.spill) {
      MM_HIP(c, c->dSketchSpill.ensure((size_t)grid * 3 * (size_t)(HTH + PADH) * 4 + 64));
      launchHard(k_sketch_hard<K, true>, c->dSketchSpill.as<int32_t>());
    } else launchHard(k_sketch_hard<K, false>, (int32_t*)nullptr);
    MM_HIP(c, hipGetLastError());
  }
  return MM_OK;
}

template <int K>
static int launch_hash_only_k(mm_ctx* c, int reps, double* msAvg) {
  using Tabs = typename MMTabsFor<K>::type;
  const int nF = (int)c->nFrags;
  const int maxLen = c->maxFragLen;
  // the sketch kernel's geometry for these fragments: positions per thread, threads per workgroup, LDS per workgroup
  const FastGeom g = sketch_fast_geom(K, c->P.sketchSize, maxLen, sizeof(Tabs), MMHasSL20<K>::value);
  const size_t need = sizeof(Tabs) + (((size_t)(maxLen + 15) / 16 + 4) * 4 + 15) / 16 * 16;
  size_t lds = (g.lds > need && g.lds <= 160 * 1024) ? g.lds : need;
  // MM_HASH_ONLY_LDS=bytes: claim that much LDS per workgroup instead (occupancy experiment); MM_HASH_ONLY_BARE=1: only what the kernel needs
  if (getenv("MM_HASH_ONLY_BARE")) lds = need;
  if (const char* e = getenv("MM_HASH_ONLY_LDS")) { const size_t want = (size_t)atol(e); if (want > need && want <= 160 * 1024) lds = want; }
  if (c->sketchTabsK != K) {
    MM_HIP(c, c->dSketchTabs.ensure(sizeof(Tabs)));
    hipLaunchKernelGGL((k_sketch_tables<K>), dim3(1), dim3(256), 0, c->stream, c->dSketchTabs.as<Tabs>());
    MM_HIP(c, hipGetLastError());
    c->sketchTabsK = K;
  }
  MM_HIP(c, c->dCounters.ensure(256));
  float total = 0;
  for (int r = 0; r <= reps; r++) {                 // launch 0 is a warm-up
    MM_HIP(c, hipEventRecord(c->evA, c->stream));
    auto launch = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, dim3(nF), dim3(g.threads), lds, c->stream, c->dSketchTabs.as<uint4>(), c->dBases2.as<uint32_t>(), c->dFrags.as<DFrag>(),
                         c->dCounters.as<uint64_t>() + 24);
    };
    if constexpr (MMHasSL20<K>::value) { if (g.SL == 20) launch(k_hash_only<K, 20>); else launch(k_hash_only<K, 16>); }
    else launch(k_hash_only<K, 16>);
    MM_HIP(c, hipGetLastError());
    MM_HIP(c, hipEventRecord(c->evB, c->stream));
    MM_HIP(c, hipEventSynchronize(c->evB));
    float ms = 0; MM_HIP(c, hipEventElapsedTime(&ms, c->evA, c->evB));
    if (r) total += ms;
  }
  *msAvg = total / reps;
  return MM_OK;
}

extern "C" int mm_bench_hash_only(mm_ctx* c, int reps, double* msAvg) {
  if (!c->nFrags || reps < 1 || !msAvg) { c->err = "mm_bench_hash_only: needs resident reads and reps >= 1"; return MM_ERR_ARG; }
  MM_HIP(c, hipSetDevice(c->device));
  switch (c->P.kmerSize) {
#define MM_CASE(KK) case KK: return launch_hash_only_k<KK>(c, reps, msAvg);
    MM_CASE(16) MM_CASE(19) MM_CASE(21)
#undef MM_CASE
    default: c->err = "mm_bench_hash_only: built for k = 16, 19, 21"; return MM_ERR_ARG;
  }
}

int mm_launch_sketch(mm_ctx* c) {
  const size_t nF = c->nFrags, s = (size_t)c->P.sketchSize;
  const size_t cF = mm_frag_cap(c, nF);               // (the largest batch the caller has announced: no reallocation when it comes)
  if (s > MM_LDS_MAX_SKETCH) {
    MM_HIP(c, c->dSkHash.ensure(nF * s * 8 + 64)); MM_HIP(c, c->dSkPos.ensure(nF * s * 8 + 64));
    MM_HIP(c, c->dSkStrand.ensure(nF * s + 64)); MM_HIP(c, c->dSkCount.ensure(nF * 4 + 64));
    MM_HIP(c, c->dHardList.ensure(nF * 4 + 64)); MM_HIP(c, c->dCounters.ensure(512));
    return mm_launch_sketch_global(c);
  }
  MM_HIP(c, c->dSkHash.ensure(cF * s * 8 + 64));
  MM_HIP(c, c->dSkPos.ensure(cF * s * 8 + 64));
  MM_HIP(c, c->dSkStrand.ensure(cF * s + 64));
  MM_HIP(c, c->dSkCount.ensure(cF * 4 + 64));
  MM_HIP(c, c->dHardList.ensure(cF * 4 + 64));
  MM_HIP(c, c->dCounters.ensure(256));
  if (nF == 0) return MM_OK;
  switch (c->P.kmerSize) {
#define MM_CASE(KK) case KK: return launch_sketch_k<KK>(c);
    MM_CASE(1) MM_CASE(2) MM_CASE(3) MM_CASE(4) MM_CASE(5) MM_CASE(6) MM_CASE(7) MM_CASE(8) MM_CASE(9) MM_CASE(10) MM_CASE(11) MM_CASE(12) MM_CASE(13) MM_CASE(14) MM_CASE(15) MM_CASE(16) MM_CASE(17) MM_CASE(18) MM_CASE(19) MM_CASE(20) MM_CASE(21) MM_CASE(22) MM_CASE(23) MM_CASE(24) MM_CASE(25) MM_CASE(26) MM_CASE(27) MM_CASE(28) MM_CASE(29) MM_CASE(30) MM_CASE(31) MM_CASE(32)
    MM_CASE(33) MM_CASE(34) MM_CASE(35) MM_CASE(36) MM_CASE(37) MM_CASE(38) MM_CASE(39) MM_CASE(40) MM_CASE(41) MM_CASE(42) MM_CASE(43) MM_CASE(44) MM_CASE(45) MM_CASE(46) MM_CASE(47) MM_CASE(48)
    MM_CASE(49) MM_CASE(50) MM_CASE(51) MM_CASE(52) MM_CASE(53) MM_CASE(54) MM_CASE(55) MM_CASE(56) MM_CASE(57) MM_CASE(58) MM_CASE(59) MM_CASE(60) MM_CASE(61) MM_CASE(62) MM_CASE(63) MM_CASE(64)
#undef MM_CASE
    default: c->err = "kmerSize outside 1..64"; return MM_ERR_ARG;
  }
}

int mm_launch_pack_raw(mm_ctx* c, const uint8_t* dAscii, const int64_t* dSrcOff, const int64_t* dPackOff, const int32_t* dLen, int nReads,
                       int64_t nChunks, uint32_t* dB, uint32_t* dM, uint32_t* dHasN) {
  if (nChunks == 0) return MM_OK;
  int blocks = (int)((nChunks + 255) / 256); if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_pack2bit, dim3(blocks), dim3(256), 0, c->stream, dAscii, dSrcOff, dPackOff, dLen, nReads, nChunks, dB, dM, dHasN);
  MM_HIP(c, hipGetLastError());
  return MM_OK;
}

int mm_launch_pack(mm_ctx* c) {
  const int64_t nChunks = (int64_t)(c->nPackedBases / 32);
  if (nChunks == 0) return MM_OK;
  KernelTimer t(c, MM_K_PACK);
  int blocks = (int)((nChunks + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_pack2bit, dim3(blocks), dim3(256), 0, c->stream, c->dAscii.as<uint8_t>(), c->dReadSrcOff.as<int64_t>(),
                     c->dReadPackOff.as<int64_t>(), c->dReadLen.as<int32_t>(), (int)c->nReads, nChunks,
                     c->dBases2.as<uint32_t>(), c->dNmask.as<uint32_t>(), c->dReadHasN.as<uint32_t>());
  MM_HIP(c, hipGetLastError());
  return MM_OK;
}

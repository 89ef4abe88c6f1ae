// mashmap_amd/csrc/mm_sketch.hip -- a1 (pack) and a4 (query-fragment sketch) kernels.
//
//   k_pack2bit          makeUpperCaseAndValidDNA            src/map/include/commonFunc.hpp:97
//   k_sketch_fragments  CommonFunc::sketchSequence          src/map/include/commonFunc.hpp:183-288
//
// One workgroup per query fragment.  Integer-ALU bound (2 x MurmurHash3_x64_128 per base); the
// packed input is only 0.25 B/bp (+0.125 B/bp N mask).  See DESIGN.md for the roofline terms.
#include "mm_internal.h"
#include "mm_device.h"
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
// LDS hash table used to de-duplicate the survivors of the threshold filter
// ---------------------------------------------------------------------------------------------
struct SkTable {
  uint64_t* key; int32_t* first; int32_t* last; int32_t* sum; uint32_t* counters;   // counters[0]=distinct, [1]=overflow
  uint32_t mask, maxLoad;
  __device__ __forceinline__ void insert(uint64_t h, int pos, int st) {
    uint32_t slot = (uint32_t)h & mask;
    while (true) {
      if (((volatile uint32_t*)counters)[1]) return;
      const unsigned long long prev = atomicCAS((unsigned long long*)&key[slot], (unsigned long long)MM_HASH_MAX,
                                                (unsigned long long)h);
      if (prev == MM_HASH_MAX) {
        if (atomicAdd(&counters[0], 1u) >= maxLoad) atomicOr(&counters[1], 1u);
      }
      if (prev == MM_HASH_MAX || prev == h) {
        atomicMin(&first[slot], pos); atomicMax(&last[slot], pos); atomicAdd(&sum[slot], st);
        return;
      }
      slot = (slot + 1) & mask;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// k_sketch_fragments<K, HARD>
//   FAST (HARD=false): hash every k-mer on both strands, keep canonical hashes below a threshold
//     T ~ 1.75 * s / (2n) * 2^64 in an LDS queue (wave ballot compaction), de-duplicate them in an LDS
//     hash table (first / last position, strand sum), bitonic-sort the distinct ones, emit the s
//     smallest.  If the queue or table overflows, or fewer than s distinct survive while T < max,
//     the fragment is appended to the hard list instead.
//   HARD: exact for any input (tandem repeats, low complexity, N-rich): survivors go straight into
//     a larger table (duplicates collapse) and T is bisected until s <= distinct <= load limit.
// ---------------------------------------------------------------------------------------------
template <int K, bool HARD>
__global__ void __launch_bounds__(1024)
k_sketch_fragments(const uint32_t* __restrict__ bases2, const uint32_t* __restrict__ nmask,
                   const DFrag* __restrict__ frags, const uint32_t* __restrict__ readHasN,
                   const int32_t* __restrict__ fragList, int s, int HT,
                   uint64_t* __restrict__ skHash, int2* __restrict__ skPos, int8_t* __restrict__ skStrand,
                   uint32_t* __restrict__ skCount, int32_t* __restrict__ hardList, uint32_t* __restrict__ hardCount) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int f = HARD ? fragList[blockIdx.x] : (int)blockIdx.x;
  const DFrag fr = frags[f];
  const int len = fr.len;
  const int n = len - K + 1;                        // k-mer positions
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (n <= 0) { if (tid == 0) skCount[f] = 0; return; }
  const bool hasN = readHasN[fr.readId] != 0;

  // ---- LDS carve (every offset a multiple of 16) ----
  const int nW = (len + 15) / 16 + 3;               // code words incl. 2 words of run-off for the last strip
  const int nM = (len + 31) / 32 + 2;
  size_t off = 0;
  MMTables* tabs = (MMTables*)(smem + off); off += sizeof(MMTables);
  mm_tables_init<K>(*tabs, tid, nthr);                           // made visible by the first __syncthreads() below
  uint32_t* sW = (uint32_t*)(smem + off); off += (((size_t)nW * 4 + 15) / 16) * 16;
  uint32_t* sM = (uint32_t*)(smem + off); off += (((size_t)nM * 4 + 15) / 16) * 16;
  uint64_t* arrA = (uint64_t*)(smem + off); off += (size_t)HT * 8;     // queue hashes, later sort keys
  uint32_t* arrB = (uint32_t*)(smem + off); off += (size_t)HT * 4;     // queue meta, later sort payload (slot)
  SkTable tab;
  tab.key = (uint64_t*)(smem + off); off += (size_t)HT * 8;
  tab.first = (int32_t*)(smem + off); off += (size_t)HT * 4;
  tab.last = (int32_t*)(smem + off); off += (size_t)HT * 4;
  tab.sum = (int32_t*)(smem + off); off += (size_t)HT * 4;
  tab.counters = (uint32_t*)(smem + off); off += 16;                   // [0] distinct [1] overflow [2] queue count [3] compact count
  tab.mask = (uint32_t)HT - 1u; tab.maxLoad = (uint32_t)HT * 5u / 8u;

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

  // threshold: expected s-th smallest of n uniform hashes is s/n * 2^64; take 1.75x (fast) as the cut
  uint64_t T;
  {
    // the canonical hash is the smaller of two uniform 64-bit values, so P[h < T] ~ 2T / 2^64
    const uint64_t want = HARD ? (uint64_t)s * 2u : ((uint64_t)s * 7u + 3u) / 4u;
    T = (want >= (uint64_t)n) ? MM_HASH_MAX : (MM_HASH_MAX / (2ull * (uint64_t)n)) * want;
  }
  uint64_t lo = 0, hi = MM_HASH_MAX; bool hiInf = true;   // HARD bisection state (uniform across the block)
  const int nStrips = (n + 15) >> 4;
  const uint64_t kmask = (K >= 64) ? ~0ull : ((1ull << K) - 1ull);
  uint32_t D = 0;

  for (int attempt = 0;; attempt++) {
    for (int i = tid; i < HT; i += nthr) { tab.key[i] = MM_HASH_MAX; tab.first[i] = 0x7fffffff; tab.last[i] = -1; tab.sum[i] = 0; }
    if (tid < 4) tab.counters[tid] = 0;
    __syncthreads();

    // ---- phase 1: hash both strands of every k-mer ----
    for (int strip = tid; strip < nStrips; strip += nthr) {
      uint64_t nm = 0;
      if (hasN) {
        const uint64_t m64 = (uint64_t)sM[strip >> 1] | ((uint64_t)sM[(strip >> 1) + 1] << 32);
        nm = m64 >> ((strip & 1) * 16);
      }
      mm_strip_hashes<K>(sW[strip], sW[strip + 1], sW[strip + 2], *tabs, [&](int j, uint64_t hf, uint64_t hr) {
        const int pos = strip * 16 + j;
        const uint64_t h = hf < hr ? hf : hr;
        bool pass = (pos < n) & (hf != hr) & (T == MM_HASH_MAX ? true : h < T);
        if (hasN) pass = pass & (((nm >> j) & kmask) == 0);
        const int sgn = hf < hr ? 1 : -1;
        if (HARD) {
          if (pass) tab.insert(h, pos, sgn);
        } else {
          const uint64_t m = __ballot(pass);
          if (m) {
            uint32_t base = 0;
            if (mm_lane() == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&tab.counters[2], (uint32_t)__popcll(m));
            base = __shfl(base, __builtin_ctzll(m));
            if (pass) {
              const uint32_t idx = base + mm_popc_below(m);
              if (idx < (uint32_t)HT) { arrA[idx] = h; arrB[idx] = ((uint32_t)pos << 1) | (sgn > 0 ? 1u : 0u); }
            }
          }
        }
      });
    }
    __syncthreads();

    bool fail = false;
    if (!HARD) {
      const uint32_t qn = tab.counters[2];
      if (qn > (uint32_t)HT) fail = true;
      else {
        for (uint32_t i = tid; i < qn; i += nthr) { const uint32_t m = arrB[i]; tab.insert(arrA[i], (int)(m >> 1), (m & 1u) ? 1 : -1); }
      }
      __syncthreads();
    }
    D = tab.counters[0];
    const bool overflow = tab.counters[1] != 0;
    if (!HARD) {
      if (fail || overflow || (D < (uint32_t)s && T != MM_HASH_MAX)) {
        if (tid == 0) { const uint32_t at = atomicAdd(hardCount, 1u); hardList[at] = f; skCount[f] = 0; }
        return;
      }
      break;
    } else {
      if (overflow) { hi = T; hiInf = false; }
      else if (D < (uint32_t)s && T != MM_HASH_MAX) { lo = T; }
      else break;
      if (hiInf) T = (T > MM_HASH_MAX / 4) ? MM_HASH_MAX : T * 4;
      else T = lo + (hi - lo) / 2;
      __syncthreads();
      if (attempt > 200) break;                     // cannot happen (bisection on 64 bits); keeps the loop bounded
    }
  }

  // ---- compact distinct entries into (key, slot) pairs, pad to a power of two, bitonic sort ----
  uint32_t n2 = 1; while (n2 < D) n2 <<= 1;
  if (n2 > (uint32_t)HT) n2 = HT;
  for (int i = tid; i < HT; i += nthr) {
    const uint64_t kx = tab.key[i];
    if (kx != MM_HASH_MAX) { const uint32_t at = atomicAdd(&tab.counters[3], 1u); arrA[at] = kx; arrB[at] = (uint32_t)i; }
  }
  __syncthreads();
  for (uint32_t i = D + tid; i < n2; i += nthr) { arrA[i] = MM_HASH_MAX; arrB[i] = 0; }
  __syncthreads();
  for (uint32_t k2 = 2; k2 <= n2; k2 <<= 1) {
    for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < n2; i += nthr) {
        const uint32_t p = i ^ j;
        if (p > i) {
          const uint64_t a = arrA[i], b = arrA[p];
          const bool up = (i & k2) == 0;
          if ((a > b) == up) { arrA[i] = b; arrA[p] = a; const uint32_t t = arrB[i]; arrB[i] = arrB[p]; arrB[p] = t; }
        }
      }
      __syncthreads();
    }
  }

  // ---- emit the s smallest distinct hashes, ascending (commonFunc.hpp:278-286) ----
  const uint32_t cnt = D < (uint32_t)s ? D : (uint32_t)s;
  for (uint32_t r = tid; r < cnt; r += nthr) {
    const uint32_t slot = arrB[r];
    const size_t o = (size_t)f * s + r;
    skHash[o] = arrA[r];
    skPos[o] = make_int2(tab.first[slot], tab.last[slot]);
    // the reference accumulates the strand in an int16 (base_types.hpp:24, commonFunc.hpp:268)
    const int16_t acc = (int16_t)tab.sum[slot];
    skStrand[o] = acc > 0 ? 1 : (acc == 0 ? 0 : -1);
  }
  if (tid == 0) skCount[f] = cnt;
}

// ---------------------------------------------------------------------------------------------
static size_t sketch_lds_bytes(int maxLen, int HT) {
  const size_t nW = (size_t)(maxLen + 15) / 16 + 3, nM = (size_t)(maxLen + 31) / 32 + 2;
  return sizeof(MMTables) + ((nW * 4 + 15) / 16) * 16 + ((nM * 4 + 15) / 16) * 16 + (size_t)HT * (8 + 4 + 8 + 4 + 4 + 4) + 16;
}
static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <int K>
static int launch_sketch_k(mm_ctx* c) {
  const int s = c->P.sketchSize;
  const int nF = (int)c->nFrags;
  const int HT = next_pow2(s * 3 < 256 ? 256 : s * 3);
  const int HTH = next_pow2(s * 4 < 4096 ? 4096 : s * 4);
  const int maxLen = c->maxFragLen;
  const size_t ldsFast = sketch_lds_bytes(maxLen, HT), ldsHard = sketch_lds_bytes(maxLen, HTH);
  if (ldsHard > 160 * 1024) { c->err = "fragment too long / sketch too large for the LDS-resident sketch kernel"; return MM_ERR_ARG; }
  int nStrips = (maxLen - K + 1 + 15) / 16; if (nStrips < 1) nStrips = 1;
  int threads = ((nStrips + 63) / 64) * 64; if (threads > 1024) threads = 1024; if (threads < 64) threads = 64;
  MM_HIP(c, hipMemsetAsync(c->dCounters.p, 0, 64, c->stream));
  MM_HIP(c, hipFuncSetAttribute((const void*)k_sketch_fragments<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsFast));
  MM_HIP(c, hipFuncSetAttribute((const void*)k_sketch_fragments<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsHard));
  {
    KernelTimer t(c, MM_K_SKETCH);
    hipLaunchKernelGGL((k_sketch_fragments<K, false>), dim3(nF), dim3(threads), ldsFast, c->stream,
                       c->dBases2.as<uint32_t>(), c->dNmask.as<uint32_t>(), c->dFrags.as<DFrag>(), c->dReadHasN.as<uint32_t>(),
                       (const int32_t*)nullptr, s, HT, c->dSkHash.as<uint64_t>(), c->dSkPos.as<int2>(), c->dSkStrand.as<int8_t>(),
                       c->dSkCount.as<uint32_t>(), c->dHardList.as<int32_t>(), c->dCounters.as<uint32_t>());
    MM_HIP(c, hipGetLastError());
  }
  uint32_t nHard = 0;
  MM_HIP(c, hipMemcpyAsync(&nHard, c->dCounters.p, 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  if (getenv("MM_DEBUG")) fprintf(stderr, "[mm] sketch: %d fragments, %u to the hard path, threads %d, HT %d, lds %zu/%zu\n", nF, nHard, threads, HT, ldsFast, ldsHard);
  if (nHard) {
    KernelTimer t(c, MM_K_SKETCH_HARD);
    hipLaunchKernelGGL((k_sketch_fragments<K, true>), dim3(nHard), dim3(threads), ldsHard, c->stream,
                       c->dBases2.as<uint32_t>(), c->dNmask.as<uint32_t>(), c->dFrags.as<DFrag>(), c->dReadHasN.as<uint32_t>(),
                       c->dHardList.as<int32_t>(), s, HTH, c->dSkHash.as<uint64_t>(), c->dSkPos.as<int2>(), c->dSkStrand.as<int8_t>(),
                       c->dSkCount.as<uint32_t>(), c->dHardList.as<int32_t>(), c->dCounters.as<uint32_t>() + 1);
    MM_HIP(c, hipGetLastError());
  }
  return MM_OK;
}

int mm_launch_sketch(mm_ctx* c) {
  const size_t nF = c->nFrags, s = (size_t)c->P.sketchSize;
  MM_HIP(c, c->dSkHash.ensure(nF * s * 8 + 64));
  MM_HIP(c, c->dSkPos.ensure(nF * s * 8 + 64));
  MM_HIP(c, c->dSkStrand.ensure(nF * s + 64));
  MM_HIP(c, c->dSkCount.ensure(nF * 4 + 64));
  MM_HIP(c, c->dHardList.ensure(nF * 4 + 64));
  MM_HIP(c, c->dCounters.ensure(256));
  if (nF == 0) return MM_OK;
  switch (c->P.kmerSize) {
#define MM_CASE(KK) case KK: return launch_sketch_k<KK>(c);
    MM_CASE(11) MM_CASE(12) MM_CASE(13) MM_CASE(14) MM_CASE(15) MM_CASE(16) MM_CASE(17) MM_CASE(18) MM_CASE(19)
    MM_CASE(20) MM_CASE(21) MM_CASE(22) MM_CASE(23) MM_CASE(24) MM_CASE(25) MM_CASE(27) MM_CASE(29) MM_CASE(31) MM_CASE(32)
#undef MM_CASE
    default: c->err = "kmerSize not compiled into the sketch kernel (supported: 11-25, 27, 29, 31, 32)"; return MM_ERR_ARG;
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

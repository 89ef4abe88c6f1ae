// mashmap_amd/csrc/mm_sketch_global.hip -- CommonFunc::sketchSequence (src/map/include/commonFunc.hpp:183-288) for sketches that no
// LDS table holds: sketchSize beyond 8 190 (the reference's --dense derives sketchSize = 0.02 (1 + (1 - pi) / 0.05) (segLength - k),
// parseCmdArgs.hpp:626-630: 9 998 for --pi 80 -s 100000, and the stock binary runs it).
//
// The order-free restatement of the reference's loop, literally and in global memory: one workgroup per fragment
//   1. the canonical hash of every k-mer position that holds no N and whose two strands differ (:207-240), with its position and strand;
//   2. a bitonic sort of (hash, position) over the fragment's slice of a scratch buffer;
//   3. one output entry per distinct hash, ascending: first = smallest position, last = largest, strand = sign of the sum of the
//      occurrences' strands in an int16 (:242-270); the s smallest (:278-286).
// Exact for any input, k 1..64 (the k-mer is hashed byte by byte at run time), any sketch size; not fast -- a fallback for parameter
// combinations far from the reference's defaults, so that they run instead of being refused.
#include "mm_internal.h"
#include "mm_device.h"
#include <algorithm>

namespace {

__device__ __forceinline__ uint32_t g_code(const uint32_t* __restrict__ bases2, int64_t i) { return (bases2[i >> 4] >> (2 * (int)(i & 15))) & 3u; }
__device__ __forceinline__ uint32_t g_ascii(uint32_t code) { return code == 0 ? 0x41u : code == 1 ? 0x43u : code == 2 ? 0x47u : 0x54u; }

// MurmurHash3_x64_128 (seed 42), low word, of the K ASCII bytes of the k-mer at base b0: forward (rc == false) or reverse complement
__device__ uint64_t g_murmur(const uint32_t* __restrict__ bases2, int64_t b0, int K, bool rc) {
  auto byteAt = [&](int i) -> uint64_t { return rc ? (uint64_t)g_ascii(3u - g_code(bases2, b0 + K - 1 - i)) : (uint64_t)g_ascii(g_code(bases2, b0 + i)); };
  uint64_t h1 = MM_SEED, h2 = MM_SEED;
  const int nb = K / 16;
  for (int b = 0; b < nb; b++) {
    uint64_t k1 = 0, k2 = 0;
    for (int i = 0; i < 8; i++) { k1 |= byteAt(16 * b + i) << (8 * i); k2 |= byteAt(16 * b + 8 + i) << (8 * i); }
    h1 ^= mm_mix_k1(k1); h1 = mm_rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    h2 ^= mm_mix_k2(k2); h2 = mm_rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const int rem = K & 15;
  uint64_t k1 = 0, k2 = 0;
  for (int i = 0; i < rem; i++) { if (i < 8) k1 |= byteAt(16 * nb + i) << (8 * i); else k2 |= byteAt(16 * nb + i) << (8 * (i - 8)); }
  if (rem > 8) h2 ^= mm_mix_k2(k2);
  if (rem > 0) h1 ^= mm_mix_k1(k1);
  h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
  h1 += h2; h2 += h1;
  h1 = mm_fmix64(h1); h2 = mm_fmix64(h2);
  return h1 + h2;
}

// one entry of the sort: the hash, then position << 1 | (forward strand is the smaller one)
struct GEnt { uint64_t h; uint32_t m; uint32_t pad; };
__device__ __forceinline__ bool g_less(const GEnt& a, const GEnt& b) { return a.h != b.h ? a.h < b.h : a.m < b.m; }

__global__ void __launch_bounds__(1024)
k_sketch_global(int nF, int K, int s, int64_t slice /* entries per workgroup, a power of two >= the longest fragment's positions */,
                const uint32_t* __restrict__ bases2, const uint32_t* __restrict__ nmask, const DFrag* __restrict__ frags, const uint32_t* __restrict__ readHasN,
                GEnt* __restrict__ scratch, uint64_t* __restrict__ skHash, int2* __restrict__ skPos, int8_t* __restrict__ skStrand, uint32_t* __restrict__ skCount) {
  __shared__ int sCount[1024];
  __shared__ int sTotal;
  GEnt* e = scratch + (size_t)blockIdx.x * (size_t)slice;
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int f = blockIdx.x; f < nF; f += gridDim.x) {
    const DFrag fr = frags[f];
    const int n = fr.len - K + 1;
    if (n <= 0) { if (tid == 0) skCount[f] = 0; continue; }
    const bool hasN = readHasN[fr.readId] != 0;
    // A cut first: only hashes below T go into the sort, T placed where s + 5 sqrt(s) + 64 of the ~2n uniform strand hashes are expected
    // (any T gives the exact sketch as long as s distinct hashes survive -- counted below; a fragment where they do not, a repeat, is done
    // again without a cut).  At sketchSize 9 998 over 100 kbp segments that is a sort of 16 384 entries instead of 131 072.
    const double want = (double)s + 5.0 * sqrt((double)s) + 64.0;
    uint64_t T = want >= (double)n ? MM_HASH_MAX : (uint64_t)(want / (2.0 * (double)n) * 18446744073709551616.0);
    int64_t N2 = 2;
    for (int attempt = 0; attempt < 2; attempt++) {
    // ---- 1. hashes below the cut, compacted (wave-aggregated cursor) ----
    if (tid == 0) sTotal = 0;
    __syncthreads();
    for (int64_t p0 = 0; p0 < n; p0 += nthr) {
      const int64_t p = p0 + tid;
      GEnt x; x.h = MM_HASH_MAX; x.m = 0xFFFFFFFFu; x.pad = 0;
      bool keep = false;
      if (p < n) {
        bool ok = true;
        if (hasN) for (int i = 0; i < K && ok; i++) { const int64_t b = fr.base + p + i; if ((nmask[b >> 5] >> (int)(b & 31)) & 1u) ok = false; }
        if (ok) {
          const uint64_t hf = g_murmur(bases2, fr.base + p, K, false), hr = g_murmur(bases2, fr.base + p, K, true);
          if (hf != hr) { x.h = hf < hr ? hf : hr; x.m = ((uint32_t)p << 1) | (hf < hr ? 1u : 0u); keep = T == MM_HASH_MAX || x.h < T; }
        }
      }
      const uint64_t m = mm_ballot(keep);
      int base = 0;
      if (m && mm_lane() == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&sTotal, (int)__popcll(m));
      if (m) base = __shfl(base, (int)__builtin_ctzll(m));
      if (keep) e[base + (int)mm_popc_below(m)] = x;
    }
    __syncthreads();
    const int nSurv = sTotal;
    N2 = 2; while (N2 < nSurv) N2 <<= 1;
    for (int64_t p = nSurv + tid; p < N2; p += nthr) { GEnt x; x.h = MM_HASH_MAX; x.m = 0xFFFFFFFFu; x.pad = 0; e[p] = x; }
    __threadfence_block();
    __syncthreads();
    // ---- 2. bitonic sort, ascending by (hash, position) ----
    for (int64_t k2 = 2; k2 <= N2; k2 <<= 1)
      for (int64_t j = k2 >> 1; j > 0; j >>= 1) {
        for (int64_t i = tid; i < N2; i += nthr) {
          const int64_t q = i ^ j;
          if (q > i) {
            const GEnt a = e[i], b = e[q];
            if (g_less(b, a) == ((i & k2) == 0)) { e[i] = b; e[q] = a; }
          }
        }
        __threadfence_block();
        __syncthreads();
      }
    // ---- 3. distinct hashes in order: every thread owns a contiguous chunk; a run belongs to the chunk its first entry lies in ----
    const int64_t chunk = (N2 + nthr - 1) / nthr;
    const int64_t c0 = (int64_t)tid * chunk, c1 = c0 + chunk < N2 ? c0 + chunk : N2;
    int mine = 0;
    for (int64_t i = c0; i < c1; i++) { const uint64_t h = e[i].h; if (h != MM_HASH_MAX && (i == 0 || e[i - 1].h != h)) mine++; }
    sCount[tid] = mine;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int t = 0; t < nthr; t++) { const int v = sCount[t]; sCount[t] = acc; acc += v; } sTotal = acc; }
    __syncthreads();
    if (sTotal < s && T != MM_HASH_MAX) { T = MM_HASH_MAX; __syncthreads(); continue; }     // fewer than s distinct hashes below the cut: once more without one
    int rank = sCount[tid];
    for (int64_t i = c0; i < c1 && rank < s; i++) {
      const uint64_t h = e[i].h;
      if (h == MM_HASH_MAX || (i != 0 && e[i - 1].h == h)) continue;
      int first = (int)(e[i].m >> 1), last = first, sum = 0;
      for (int64_t r = i; r < N2 && e[r].h == h; r++) { last = (int)(e[r].m >> 1); sum += (e[r].m & 1u) ? 1 : -1; }
      const size_t o = (size_t)f * s + rank;
      skHash[o] = h; skPos[o] = make_int2(first, last);
      const int16_t acc = (int16_t)sum;                                    // the reference accumulates the strand in an int16 (base_types.hpp:24)
      skStrand[o] = acc > 0 ? 1 : (acc == 0 ? 0 : -1);
      rank++;
    }
    if (tid == 0) skCount[f] = (uint32_t)(sTotal < s ? sTotal : s);
    __syncthreads();
    break;
    }  // attempt
  }
}

}  // namespace

// every resident fragment through the global-memory sketch (mm_launch_sketch routes here when the sketch does not fit the LDS kernels)
int mm_launch_sketch_global(mm_ctx* c) {
  const int nF = (int)c->nFrags, K = c->P.kmerSize, s = c->P.sketchSize;
  if (nF == 0) return MM_OK;
  int64_t slice = 2; while (slice < c->maxFragLen) slice <<= 1;
  // as many workgroups as ~8 GiB of scratch allow, at most 1024 (a CU runs two of them at a time)
  int grid = (int)std::min<int64_t>(std::min<int64_t>(nF, 1024), std::max<int64_t>(1, ((int64_t)8 << 30) / (slice * (int64_t)sizeof(GEnt))));
  MM_HIP(c, c->dSketchSpill.ensure((size_t)grid * (size_t)slice * sizeof(GEnt) + 64));
  KernelTimer t(c, MM_K_SKETCH_HARD);
  hipLaunchKernelGGL(k_sketch_global, dim3(grid), dim3(1024), 0, c->stream, nF, K, s, slice, c->dBases2.as<uint32_t>(), c->dNmask.as<uint32_t>(), c->dFrags.as<DFrag>(),
                     c->dReadHasN.as<uint32_t>(), c->dSketchSpill.as<GEnt>(), c->dSkHash.as<uint64_t>(), c->dSkPos.as<int2>(), c->dSkStrand.as<int8_t>(),
                     c->dSkCount.as<uint32_t>());
  MM_HIP(c, hipGetLastError());
  return MM_OK;
}

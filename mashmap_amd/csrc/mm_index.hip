// mashmap_amd/csrc/mm_index.hip -- reference-side index construction (mm_index_build).
//
//   k_ref_hash        both-strand MurmurHash3 of every reference k-mer        commonFunc.hpp:357-373
//   (mm_winnow.hip)   sliding bottom-s "minmer" intervals of one contig       commonFunc.hpp:302-521
//   finish_contig     tail of addMinmers: drop empty runs, chunk, sort, unique commonFunc.hpp:523-568
//   build_lookup      Sketch::index                                           winSketch.hpp:379-404
//   frequency_filter  computeFreqHist / computeFreqSeedSet / dropFreqSeedSet  winSketch.hpp:410-504
//
// Hashing and winnowing run on the device; the host stitches the tiles' open runs together and applies the reference's own
// tail (std::sort + std::unique on the emission order, which is what fixes the order of ties) and Sketch::index.
#include "mm_internal.h"
#include "mm_device.h"
#include "mm_winnow.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <future>
#include <memory>
#include <thread>
#include <map>
#include <numeric>
#include <queue>
#include <unordered_map>

// ---------------------------------------------------------------------------------------------
// device: canonical hash + strand of every k-mer position of a packed contig
//   outH[i] = min(fwd, rc)  (0xFFFF... when the k-mer holds an N or fwd == rc), outS[i] = fwd < rc ? +1 : -1
// ---------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256)
k_ref_hash(const uint32_t* __restrict__ bases2, const uint32_t* __restrict__ nmask, int64_t nPos, int hasN,
           uint64_t* __restrict__ outH, int8_t* __restrict__ outS) {
  __shared__ MMTables tabs;
  mm_tables_init<K>(tabs, threadIdx.x, blockDim.x);
  __syncthreads();
  const int64_t nStrips = (nPos + 15) >> 4;
  const uint64_t kmask = K >= 64 ? ~0ull : (1ull << (K & 63)) - 1ull;
  for (int64_t strip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; strip < nStrips; strip += (int64_t)gridDim.x * blockDim.x) {
    uint64_t nm = 0; uint32_t bad = 0;                              // bad (k-mers of more than 32 bases): bit j = position j's k-mer holds an N
    if (hasN) {
      const uint64_t m64 = (uint64_t)nmask[strip >> 1] | ((uint64_t)nmask[(strip >> 1) + 1] << 32);
      nm = m64 >> ((strip & 1) * 16);
      if constexpr (MMWideK<K>::value) {
        const uint64_t h64 = (uint64_t)nmask[(strip >> 1) + 2] | ((uint64_t)nmask[(strip >> 1) + 3] << 32);
        const int sh = (int)(strip & 1) * 16;
        bad = mm_window_or_wide<K>(sh ? ((m64 >> sh) | (h64 << (64 - sh))) : m64, sh ? (h64 >> sh) : h64);
      }
    }
    uint64_t hs[16]; uint32_t sbits = 0;
    auto onPos = [&](int j, uint64_t hf, uint64_t hr) {
      bool ok = hf != hr;
      if (hasN) ok = ok & (MMWideK<K>::value ? ((bad >> j) & 1u) == 0u : ((nm >> j) & kmask) == 0);
      hs[j] = ok ? (hf < hr ? hf : hr) : MM_HASH_MAX;
      sbits |= (hf < hr ? 1u : 0u) << j;
    };
    if constexpr (MMWideK<K>::value) {
      uint32_t ww[MMWideK<K>::NW];
#pragma unroll
      for (int i = 0; i < MMWideK<K>::NW; i++) ww[i] = bases2[strip + i];
      mm_strip_hashes_wide<K>(ww, tabs, onPos);
    } else mm_strip_hashes<K>(bases2[strip], bases2[strip + 1], bases2[strip + 2], tabs, onPos);
    const int64_t p0 = strip * 16;
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (p0 + j < nPos) { outH[p0 + j] = hs[j]; outS[p0 + j] = ((sbits >> j) & 1u) ? 1 : -1; }
  }
}


// ---------------------------------------------------------------------------------------------
// host: MurmurHash3_x64_128 low word over arbitrary bytes (only for the <= k-1 leading k-mers of a
// contig that contain an 'N' the reference does not notice: commonFunc.hpp:334 has no initial-N scan)
// ---------------------------------------------------------------------------------------------
static inline uint64_t h_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t h_fmix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }
static uint64_t host_murmur(const unsigned char* p, int len) {
  uint64_t h1 = MM_SEED, h2 = MM_SEED;
  const int nb = len / 16;
  for (int b = 0; b < nb; b++) {
    uint64_t k1, k2; std::memcpy(&k1, p + 16 * b, 8); std::memcpy(&k2, p + 16 * b + 8, 8);
    k1 *= MM_C1; k1 = h_rotl(k1, 31); k1 *= MM_C2; h1 ^= k1; h1 = h_rotl(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= MM_C2; k2 = h_rotl(k2, 33); k2 *= MM_C1; h2 ^= k2; h2 = h_rotl(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  uint64_t k1 = 0, k2 = 0; const unsigned char* t = p + 16 * nb; const int rem = len & 15;
  for (int i = 0; i < rem; i++) { if (i < 8) k1 |= (uint64_t)t[i] << (8 * i); else k2 |= (uint64_t)t[i] << (8 * (i - 8)); }
  if (rem > 8) { k2 *= MM_C2; k2 = h_rotl(k2, 33); k2 *= MM_C1; h2 ^= k2; }
  if (rem > 0) { k1 *= MM_C1; k1 = h_rotl(k1, 31); k1 *= MM_C2; h1 ^= k1; }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len; h1 += h2; h2 += h1; h1 = h_fmix(h1); h2 = h_fmix(h2);
  return h1 + h2;
}

// ---------------------------------------------------------------------------------------------
// host: the device's records of one contig (reference emission order; runs that were open at a tile boundary still carry
// WN_CARRY as their start) -> the contig's slice of minmerIndex.  Stitching, then the literal tail of addMinmers
// (commonFunc.hpp:523-568): drop empty / negative runs, FWD for a zero strand sum, split runs longer than w, std::sort by
// (wpos, wpos_end), std::unique on (wpos, hash).
// ---------------------------------------------------------------------------------------------
static void finish_contig(std::vector<mm_minmer>& out, const std::vector<int32_t>& tileCount, const std::vector<WnOpenRun>& openRuns,
                          const std::vector<int32_t>& openCount, int s, int w, int seqId) {
  {
    std::vector<std::pair<uint64_t, int32_t>> prev, cur;                 // (hash, resolved start) of the runs open at the last boundary
    auto lookup = [&prev](uint64_t h) {
      auto it = std::lower_bound(prev.begin(), prev.end(), std::make_pair(h, (int32_t)0x80000000));
      return (it != prev.end() && it->first == h) ? it->second : (int32_t)0;   // always found: both tiles hold the sketch of that window
    };
    size_t off = 0;
    for (size_t t = 0; t < tileCount.size(); t++) {
      for (size_t i = off; i < off + (size_t)tileCount[t]; i++) if (out[i].wpos == WN_CARRY) out[i].wpos = lookup(out[i].hash);
      off += (size_t)tileCount[t];
      cur.clear();
      for (int j = 0; j < openCount[t]; j++) {
        const WnOpenRun& o = openRuns[t * (size_t)s + j];
        cur.emplace_back(o.hash, o.start == WN_CARRY ? lookup(o.hash) : o.start);
      }
      std::sort(cur.begin(), cur.end());
      prev.swap(cur);
    }
  }
  for (auto& m : out) m.seqId = seqId;
  out.erase(std::remove_if(out.begin(), out.end(), [](const mm_minmer& m) { return m.wpos < 0 || m.wpos_end < 0 || m.wpos == m.wpos_end; }), out.end());
  std::vector<mm_minmer> pieces;
  for (auto& m : out) {
    m.strand = m.strand < 0 ? -1 : 1;
    if (m.wpos_end > m.wpos + w) {
      const int nchunk = (int)std::ceil(float(m.wpos_end - m.wpos) / float(w));
      for (int c = 0; c < nchunk; c++) pieces.push_back(mm_minmer{m.hash, m.wpos + c * w, std::min(m.wpos + c * w + w, m.wpos_end), m.seqId, m.strand, 0});
    }
  }
  out.erase(std::remove_if(out.begin(), out.end(), [w](const mm_minmer& m) { return m.wpos_end - m.wpos > w; }), out.end());
  out.insert(out.end(), pieces.begin(), pieces.end());
  std::sort(out.begin(), out.end(), [](const mm_minmer& l, const mm_minmer& r) { return l.wpos != r.wpos ? l.wpos < r.wpos : l.wpos_end < r.wpos_end; });
  out.erase(std::unique(out.begin(), out.end(), [](const mm_minmer& l, const mm_minmer& r) { return l.wpos == r.wpos && l.hash == r.hash; }), out.end());
}

// ---------------------------------------------------------------------------------------------
template <int K>
static int hash_contig(mm_ctx* c, const char* seq, int len, DevBuf& dAscii, DevBuf& dB, DevBuf& dM, DevBuf& dMeta, DevBuf& dH, DevBuf& dS) {
  const int64_t nPos = (int64_t)len - K + 1;
  const int64_t packed = ((int64_t)len + 31) / 32 * 32;
  MM_HIP(c, dAscii.ensure((size_t)len + 64)); MM_HIP(c, dB.ensure((size_t)packed / 4 + 64)); MM_HIP(c, dM.ensure((size_t)packed / 8 + 64));
  MM_HIP(c, dMeta.ensure(64)); MM_HIP(c, dH.ensure((size_t)nPos * 8 + 64)); MM_HIP(c, dS.ensure((size_t)nPos + 64));
  int64_t meta[4] = {0, packed, 0, packed};                       // srcOff[0..1], packOff[0..1]
  int32_t rl = len; uint32_t zero = 0;
  MM_HIP(c, hipMemcpyAsync(dAscii.p, seq, (size_t)len, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemcpyAsync(dMeta.p, meta, 32, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemcpyAsync((char*)dMeta.p + 32, &rl, 4, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemcpyAsync((char*)dMeta.p + 40, &zero, 4, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemsetAsync((char*)dB.p + packed / 4, 0, 64, c->stream));
  MM_HIP(c, hipMemsetAsync((char*)dM.p + packed / 8, 0, 64, c->stream));
  const int64_t nChunks = packed / 32;
  {
    const int prc = mm_launch_pack_raw(c, dAscii.as<uint8_t>(), dMeta.as<int64_t>(), dMeta.as<int64_t>() + 2, (const int32_t*)((char*)dMeta.p + 32), 1,
                                       nChunks, dB.as<uint32_t>(), dM.as<uint32_t>(), (uint32_t*)((char*)dMeta.p + 40));
    if (prc != MM_OK) return prc;
  }
  uint32_t hasN = 0;
  MM_HIP(c, hipMemcpyAsync(&hasN, (char*)dMeta.p + 40, 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  {
    KernelTimer t(c, MM_K_REFHASH);
    const int64_t nStrips = (nPos + 15) / 16;
    int b2 = (int)std::min<int64_t>((nStrips + 255) / 256, 65536);
    hipLaunchKernelGGL((k_ref_hash<K>), dim3(b2), dim3(256), 0, c->stream, dB.as<uint32_t>(), dM.as<uint32_t>(), nPos, (int)hasN,
                       dH.as<uint64_t>(), dS.as<int8_t>());
    MM_HIP(c, hipGetLastError());
  }
  // leading k-mers with an unnoticed N (see host_murmur): an N at position p < K-1 never starts the reference's
  // ambiguity countdown, so k-mers i <= p that have no N at a position >= K-1 are hashed with the 'N' byte in place.
  // At most K-1 values per contig: computed here and patched into the device arrays.
  if (hasN) {
    std::vector<unsigned char> norm((size_t)std::min<int64_t>(len, 2 * K));
    for (size_t j = 0; j < norm.size(); j++) {
      unsigned char ch = (unsigned char)seq[j]; if (ch > 96 && ch < 123) ch -= 32;
      norm[j] = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? ch : 'N';
    }
    for (int i = 0; i < K - 1 && i < nPos; i++) {
      bool early = false, late = false;
      for (int p = i; p < i + K; p++) if (norm[p] == 'N') { if (p < K - 1) early = true; else late = true; }
      if (early && !late) {
        unsigned char rc[64];
        for (int p = 0; p < K; p++) { unsigned char ch = norm[i + p]; rc[K - 1 - p] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : ch; }
        const uint64_t f = host_murmur(&norm[i], K), b = host_murmur(rc, K);
        const uint64_t hv = f == b ? MM_HASH_MAX : std::min(f, b); const int8_t sv = f < b ? 1 : -1;
        MM_HIP(c, hipMemcpyAsync((uint64_t*)dH.p + i, &hv, 8, hipMemcpyHostToDevice, c->stream));
        MM_HIP(c, hipMemcpyAsync((int8_t*)dS.p + i, &sv, 1, hipMemcpyHostToDevice, c->stream));
        MM_HIP(c, hipStreamSynchronize(c->stream));
      }
    }
  }
  return MM_OK;
}

typedef int (*HashContigFn)(mm_ctx*, const char*, int, DevBuf&, DevBuf&, DevBuf&, DevBuf&, DevBuf&, DevBuf&);
static HashContigFn pick_hasher(int k) {
  switch (k) {
#define MM_CASE(KK) case KK: return &hash_contig<KK>;
    MM_CASE(1) MM_CASE(2) MM_CASE(3) MM_CASE(4) MM_CASE(5) MM_CASE(6) MM_CASE(7) MM_CASE(8) MM_CASE(9) MM_CASE(10) MM_CASE(11) MM_CASE(12) MM_CASE(13) MM_CASE(14) MM_CASE(15) MM_CASE(16) MM_CASE(17) MM_CASE(18) MM_CASE(19) MM_CASE(20) MM_CASE(21) MM_CASE(22) MM_CASE(23) MM_CASE(24) MM_CASE(25) MM_CASE(26) MM_CASE(27) MM_CASE(28) MM_CASE(29) MM_CASE(30) MM_CASE(31) MM_CASE(32)
    MM_CASE(33) MM_CASE(34) MM_CASE(35) MM_CASE(36) MM_CASE(37) MM_CASE(38) MM_CASE(39) MM_CASE(40) MM_CASE(41) MM_CASE(42) MM_CASE(43) MM_CASE(44) MM_CASE(45) MM_CASE(46) MM_CASE(47) MM_CASE(48)
    MM_CASE(49) MM_CASE(50) MM_CASE(51) MM_CASE(52) MM_CASE(53) MM_CASE(54) MM_CASE(55) MM_CASE(56) MM_CASE(57) MM_CASE(58) MM_CASE(59) MM_CASE(60) MM_CASE(61) MM_CASE(62) MM_CASE(63) MM_CASE(64)
#undef MM_CASE
    default: return nullptr;
  }
}

// ---------------------------------------------------------------------------------------------
// Second half of the index build, shared by mm_index_build and mm_index_upload_full (--loadIndex):
//   Sketch::index (winSketch.hpp:379-404) unless the lookup map is supplied, computeFreqHist / computeFreqSeedSet /
//   dropFreqSeedSet (:410-504), then the flat device index.  `all` is minmerIndex BEFORE the frequent-seed drop.
// ---------------------------------------------------------------------------------------------
// mm_index_build: Sketch::index, the frequency filter and the flat index on the device (mm_index_dev.hip), from the contigs' record
// arrays as they are; the host copies behind mm_index_download are filled from the device on request
static int finalize_built_index(mm_ctx* c, std::vector<std::vector<mm_minmer>>& per, float kmerPctThreshold, const int32_t* contigLen,
                                const int32_t* refGroup, size_t nContigs) {
  c->hKeys.clear(); c->hOffsets.clear(); c->hPoints.clear(); c->hFreq.clear(); c->hMinmers.clear();
  c->mirrorMinmers = c->mirrorMap = false;
  std::vector<std::pair<const mm_minmer*, size_t>> parts;
  for (const auto& v : per) parts.emplace_back(v.data(), v.size());
  const int rc = mm_finalize_index_device(c, parts, kmerPctThreshold, contigLen, refGroup, nContigs);
  std::vector<mm_minmer>().swap(c->hMinmersAll);
  if (c->keepFullIndex) for (const auto& v : per) c->hMinmersAll.insert(c->hMinmersAll.end(), v.begin(), v.end());   // --saveIndex wants minmerIndex before the drop
  return rc;
}

// --loadIndex: second half of the index build from host arrays in reference layout
static int finalize_index(mm_ctx* c, std::vector<mm_minmer>& all, bool haveMap, float kmerPctThreshold, const int32_t* contigLen,
                          const int32_t* refGroup, size_t nContigs) {
  (void)haveMap;
  // --loadIndex: the lookup map comes from the file (it is not re-derived), so the threshold is taken from it here, as
  // computeFreqHist / computeFreqSeedSet / dropFreqSeedSet do (winSketch.hpp:410-504)
  int32_t freqThreshold = 0x7fffffff;
  const size_t nKeysAll = c->hKeys.size();
  if (nKeysAll) {
    std::map<int, int64_t> hist;
    for (size_t i = 0; i < nKeysAll; i++) hist[(int)(c->hOffsets[i + 1] - c->hOffsets[i])] += 1;
    const int64_t total = (int64_t)nKeysAll;
    const int64_t toIgnore = (int64_t)(total * kmerPctThreshold / 100);      // int64 * float / int, as winSketch.hpp:425
    int64_t sum = 0;
    for (auto it = hist.rbegin(); it != hist.rend(); ++it) {
      sum += it->second;
      if (sum < toIgnore) freqThreshold = it->first;
      else if (sum == toIgnore) { freqThreshold = it->first; break; }
      else break;
    }
  }
  for (size_t i = 0; i < nKeysAll; i++)
    if ((int64_t)(c->hOffsets[i + 1] - c->hOffsets[i]) >= (int64_t)freqThreshold) c->hFreq.push_back(c->hKeys[i]);
  if (c->keepFullIndex) c->hMinmersAll = all; else std::vector<mm_minmer>().swap(c->hMinmersAll);
  if (!c->hFreq.empty())
    all.erase(std::remove_if(all.begin(), all.end(), [&](const mm_minmer& m) { return std::binary_search(c->hFreq.begin(), c->hFreq.end(), m.hash); }), all.end());
  c->hMinmers.swap(all);
  c->freqThreshold = freqThreshold;
  c->mapped = false;
  return mm_build_device_index(c, contigLen, refGroup, nContigs);
}

extern "C" int mm_index_build(mm_ctx* c, const char* bases, const int64_t* contigOffsets, size_t nContigs, const int32_t* refGroup,
                              float kmerPctThreshold) {
  if (!bases || !contigOffsets || !nContigs) { c->err = "mm_index_build: null argument"; return MM_ERR_ARG; }
  MM_HIP(c, hipSetDevice(c->device));
  const int k = c->P.kmerSize, w = c->P.segLength, s = c->P.sketchSize;
  HashContigFn hasher = pick_hasher(k);
  if (!hasher) { c->err = "mm_index_build: kmerSize not compiled in"; return MM_ERR_ARG; }
  const bool dbg = getenv("MM_DEBUG") != nullptr;
  const auto tStart = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count(); };
  std::vector<int32_t> clen(nContigs);
  std::vector<std::vector<mm_minmer>> per(nContigs);
  std::deque<std::shared_future<void>> inflight;
  std::shared_future<void> stagedBusy[2];               // the finishing job that still reads the page-locked landing buffer of that turn
  DevBuf dAscii, dB, dM, dMeta, dH, dS;
  WinnowBuffers wb;
  int rc = MM_OK;
  const unsigned maxJobs = std::max(1u, std::thread::hardware_concurrency());
  for (size_t ci = 0; ci < nContigs && rc == MM_OK; ci++) {
    const int64_t len64 = contigOffsets[ci + 1] - contigOffsets[ci];
    if (len64 < 0 || len64 > 0x7fffffff) { c->err = "mm_index_build: contig longer than int32 (offset_t)"; rc = MM_ERR_ARG; break; }
    const int len = (int)len64; clen[ci] = len;
    if (len < k || len < w) continue;            // winSketch.hpp:194; contigs shorter than a window yield no minmers
    rc = hasher(c, bases + contigOffsets[ci], len, dAscii, dB, dM, dMeta, dH, dS);
    if (rc != MM_OK) break;
    auto rec = std::make_shared<std::vector<mm_minmer>>(); auto tc = std::make_shared<std::vector<int32_t>>();
    auto runs = std::make_shared<std::vector<WnOpenRun>>(); auto oc = std::make_shared<std::vector<int32_t>>();
    const int turn = wb.turn;
    if (stagedBusy[turn].valid()) stagedBusy[turn].wait();        // contig ci-2 has copied its records out of this landing buffer
    WnStaged staged;
    rc = mm_winnow_contig_device(c, wb, dH.as<uint64_t>(), dS.as<int8_t>(), (int64_t)len - k + 1, len, *rec, *tc, *runs, *oc, &staged);
    if (rc != MM_OK) break;
    while (inflight.size() >= maxJobs) { inflight.front().get(); inflight.pop_front(); }
    std::vector<mm_minmer>* dst = &per[ci];
    auto copiedOut = std::make_shared<std::promise<void>>();
    std::shared_future<void> copiedOutF = copiedOut->get_future().share();
    inflight.push_back(std::async(std::launch::async, [rec, tc, runs, oc, w, s, ci, dst, staged, copiedOut]() {
      staged.take(*rec, *runs);                                     // out of the page-locked landing buffer, off the device's critical path
      copiedOut->set_value();                                       // the buffer may take the contig after next from here on
      finish_contig(*rec, *tc, *runs, *oc, s, w, (int)ci);
      dst->swap(*rec);
    }).share());
    if (staged.hs) { stagedBusy[turn] = copiedOutF; wb.turn ^= 1; }
  }
  if (dbg) fprintf(stderr, "[mm] index: hash + winnow of %zu contigs issued at %.2f s\n", nContigs, since());
  while (!inflight.empty()) { inflight.front().get(); inflight.pop_front(); }
  if (dbg) fprintf(stderr, "[mm] index: host tails (stitch, sort, unique) done at %.2f s\n", since());
  dAscii.release(); dB.release(); dM.release(); dMeta.release(); dH.release(); dS.release(); wb.release();
  if (rc != MM_OK) return rc;

  size_t nRecords = 0; for (const auto& v : per) nRecords += v.size();
  if (dbg) fprintf(stderr, "[mm] index: %zu records in %zu per-contig arrays at %.2f s\n", nRecords, nContigs, since());
  const int frc = finalize_built_index(c, per, kmerPctThreshold, clen.data(), refGroup, nContigs);
  if (dbg) fprintf(stderr, "[mm] index: Sketch::index + frequency filter + flat index on the device done at %.2f s (%zu keys, %zu points)\n", since(), c->idx.nKeys, c->idx.nPoints);
  if (c->profile && frc == MM_OK) { MM_HIP(c, hipStreamSynchronize(c->stream)); mm_profile_collect(c); }   // the build's brackets (hash + winnow per contig) go back to the pool
  return frc;
}

extern "C" int mm_index_upload_full(mm_ctx* c, const mm_minmer* minmersAll, size_t nMinmers, const uint64_t* keys, const uint64_t* offsets,
                                    size_t nKeys, const mm_interval_point* points, size_t nPoints, const int32_t* contigLen,
                                    const int32_t* refGroup, size_t nContigs, float kmerPctThreshold) {
  if ((nMinmers && !minmersAll) || (nKeys && (!keys || !offsets)) || (nPoints && !points) || !contigLen || !nContigs) {
    c->err = "mm_index_upload_full: null argument"; return MM_ERR_ARG;
  }
  if (nKeys && offsets[nKeys] != nPoints) { c->err = "mm_index_upload_full: offsets[nKeys] != nPoints"; return MM_ERR_ARG; }
  MM_HIP(c, hipSetDevice(c->device));
  std::vector<mm_minmer> all(minmersAll, minmersAll + nMinmers);
  // the saved map comes in the saving program's iteration order: put the keys in ascending order (what mm_index_download promises)
  std::vector<uint32_t> ord(nKeys);
  std::iota(ord.begin(), ord.end(), 0u);
  std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return keys[x] < keys[y]; });
  c->hKeys.clear(); c->hOffsets.clear(); c->hPoints.clear(); c->hFreq.clear();
  c->hKeys.reserve(nKeys); c->hOffsets.reserve(nKeys + 1); c->hPoints.reserve(nPoints);
  for (size_t i = 0; i < nKeys; i++) {
    const uint32_t j = ord[i];
    if (i && keys[j] == c->hKeys.back()) { c->err = "mm_index_upload_full: duplicate key"; return MM_ERR_ARG; }
    c->hKeys.push_back(keys[j]); c->hOffsets.push_back((uint64_t)c->hPoints.size());
    c->hPoints.insert(c->hPoints.end(), points + offsets[j], points + offsets[j + 1]);
  }
  c->hOffsets.push_back((uint64_t)c->hPoints.size());
  return finalize_index(c, all, true, kmerPctThreshold, contigLen, refGroup, nContigs);
}

extern "C" int mm_index_download_full(mm_ctx* c, mm_minmer* out, size_t* n) {
  if (!c->idx.ready || !c->keepFullIndex) { c->err = "mm_index_download_full: needs MM_OPT_KEEP_FULL_INDEX before the index is built"; return MM_ERR_STATE; }
  if (n) *n = c->hMinmersAll.size();
  if (out && !c->hMinmersAll.empty()) std::memcpy(out, c->hMinmersAll.data(), c->hMinmersAll.size() * sizeof(mm_minmer));
  return MM_OK;
}


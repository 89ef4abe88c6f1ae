// mashmap_amd/csrc/mm_index_dev.hip -- the reference index after addMinmers, built on the device (gfx950):
//
//   Sketch::index           hash -> interval points, adjacent runs merged      src/map/include/winSketch.hpp:379-404
//   computeFreqHist         threshold on the number of points per hash          winSketch.hpp:410-453
//   computeFreqSeedSet      hashes at or above it                               winSketch.hpp:488-495
//   dropFreqSeedSet         minmerIndex without them                            winSketch.hpp:497-504
//   + the flat device index Map reads: merged event stream, per-block open-record lists, seed hash table,
//     presence bitmap, packed interval points
//
// The reference builds an insertion-ordered hash map with one serial pass over minmerIndex.  The data-parallel restatement:
// grouping the records by hash with the records of a hash in minmerIndex order is a STABLE sort of (hash, index) pairs (rocPRIM
// radix sort: the sort primitive only); the map's "the previous point of this hash ends where this record starts" test (:388-396)
// then looks at the neighbour in sorted order, so runs of merged records are flagged element-wise and counted with a scan.  The
// frequency threshold (:424-441: walk the point-count histogram from the top until `minmerToIgnore` hashes are covered) is a rank
// statistic of the sorted counts.  The event stream is a stable sort of 2n (contig, pos*2 + isInsert) keys; open-record lists are
// one wave per 1024-position block scanning the <= segLength positions of records before the boundary.
// Host work left: copying records in, a handful of scalars out.
#include "mm_internal.h"
#include "mm_device.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define MM_EMPTY MM_HT_EMPTY

#define K_LAUNCH(kern, n, ...) hipLaunchKernelGGL(kern, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, c->stream, __VA_ARGS__)

// ---------------------------------------------------------------------------------------------
// Sketch::index
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_idx_keys(size_t n, const mm_minmer* __restrict__ rec, uint64_t* __restrict__ h, uint32_t* __restrict__ idx) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { h[i] = rec[i].hash; idx[i] = (uint32_t)i; }
}

// per sorted position j: bit 0 = first record of a run of merged records, bit 32 = first record of its hash
__global__ void __launch_bounds__(256)
k_idx_flags(size_t n, const mm_minmer* __restrict__ rec, const uint64_t* __restrict__ sh, const uint32_t* __restrict__ sidx, uint64_t* __restrict__ flags,
            unsigned long long* __restrict__ diag /* [0] |= 1: sort not stable */) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  bool keyStart = j == 0 || sh[j] != sh[j - 1];
  bool runStart = keyStart;
  if (!keyStart) {
    if (sidx[j] < sidx[j - 1]) atomicOr(&diag[0], 1ull);                       // records of a hash must stay in minmerIndex order
    runStart = rec[sidx[j - 1]].wpos_end != rec[sidx[j]].wpos;                 // winSketch.hpp:388: back().pos != mi.wpos -> a new OPEN/CLOSE pair
  }
  flags[j] = (runStart ? 1ull : 0ull) | (keyStart ? (1ull << 32) : 0ull);
}

// OPEN point of every run, key table; pre[] = exclusive scan of flags (low word: runs before j, high word: keys before j)
__global__ void __launch_bounds__(256)
k_idx_open(size_t n, const mm_minmer* __restrict__ rec, const uint64_t* __restrict__ sh, const uint32_t* __restrict__ sidx, const uint64_t* __restrict__ flags,
           const uint64_t* __restrict__ pre, uint64_t* __restrict__ keys, uint64_t* __restrict__ keyOff, uint64_t* __restrict__ pt, int32_t* __restrict__ runSeq) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint64_t f = flags[j];
  const uint64_t run = pre[j] & 0xffffffffull, key = pre[j] >> 32;
  if (f & 1ull) {
    const mm_minmer m = rec[sidx[j]];
    pt[2 * run] = ((uint64_t)(uint32_t)m.seqId << 33) | ((uint64_t)(uint32_t)m.wpos << 1) | 1ull;
    runSeq[run] = m.seqId;
  }
  if (f >> 32) { keys[key] = sh[j]; keyOff[key] = 2 * run; }
}
// CLOSE point of every run: at the wpos_end of its last record, with the seqId of its FIRST record (the merge only moves `pos`, :395)
__global__ void __launch_bounds__(256)
k_idx_close(size_t n, const mm_minmer* __restrict__ rec, const uint32_t* __restrict__ sidx, const uint64_t* __restrict__ flags, const uint64_t* __restrict__ pre,
            const int32_t* __restrict__ runSeq, uint64_t* __restrict__ pt) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  if (j + 1 < n && !(flags[j + 1] & 1ull)) return;                             // not the last record of its run
  const uint64_t run = (pre[j] & 0xffffffffull) + (flags[j] & 1ull) - 1ull;
  pt[2 * run + 1] = ((uint64_t)(uint32_t)runSeq[run] << 33) | ((uint64_t)(uint32_t)rec[sidx[j]].wpos_end << 1);
}

// ---------------------------------------------------------------------------------------------
// frequency filter
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_key_counts(size_t nk, const uint64_t* __restrict__ keyOff, uint32_t* __restrict__ cnt) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k < nk) cnt[k] = (uint32_t)(keyOff[k + 1] - keyOff[k]);
}
// computeFreqHist (:424-441) on the ascending counts A[0, nk): walking the histogram from the largest count down, the threshold is
// the smallest count v such that at most T hashes have >= v points; none (T == 0, or ties at the T-th largest reach further) -> INT_MAX
__global__ void k_freq_threshold(size_t nk, const uint32_t* __restrict__ A, unsigned long long T, int32_t* __restrict__ thr) {
  if (threadIdx.x || blockIdx.x) return;
  int32_t out = 0x7fffffff;
  if (T >= 1 && T <= nk) {
    const uint32_t x = A[nk - T];
    if (nk - T == 0 || A[nk - T - 1] < x) out = (int32_t)x;
    else {
      size_t lo = nk - T, hi = nk;                                              // first index with A > x
      while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (A[mid] > x) hi = mid; else lo = mid + 1; }
      if (lo < nk) out = (int32_t)A[lo];
    }
  }
  *thr = out;
}
__global__ void __launch_bounds__(256)
k_key_freq(size_t nk, const uint64_t* __restrict__ keyOff, const int32_t* __restrict__ thr, uint8_t* __restrict__ freq, unsigned long long* __restrict__ nFreq) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nk) return;
  const bool f = (long long)(keyOff[k + 1] - keyOff[k]) >= (long long)*thr;     // computeFreqSeedSet (:488-495)
  freq[k] = f ? 1 : 0;
  if (f) atomicAdd(nFreq, 1ull);
}
__global__ void __launch_bounds__(256)
k_freq_list(size_t nk, const uint64_t* __restrict__ keys, const uint8_t* __restrict__ freq, uint64_t* __restrict__ out, unsigned long long* __restrict__ cursor) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k < nk && freq[k]) out[atomicAdd(cursor, 1ull)] = keys[k];
}
// dropFreqSeedSet (:497-504): keep[i] = the record's hash is not frequent (i in minmerIndex order)
__global__ void __launch_bounds__(256)
k_keep_flags(size_t n, const uint32_t* __restrict__ sidx, const uint64_t* __restrict__ flags, const uint64_t* __restrict__ pre, const uint8_t* __restrict__ freq,
             int32_t* __restrict__ keep) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint64_t key = (pre[j] >> 32) + (flags[j] >> 32) - 1ull;
  keep[sidx[j]] = freq[key] ? 0 : 1;
}
__global__ void __launch_bounds__(256)
k_compact_records(size_t n, const mm_minmer* __restrict__ rec, const int32_t* __restrict__ keep, const int64_t* __restrict__ pos, mm_minmer* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && keep[i]) out[pos[i]] = rec[i];
}

// ---------------------------------------------------------------------------------------------
// flat device index
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_check_records(size_t n, const mm_minmer* __restrict__ rec, int nContigs, unsigned long long* __restrict__ diag /* [1] bad order/position, [2] max length */) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const mm_minmer m = rec[i];
  if (m.wpos < 0 || m.wpos_end < 0 || m.seqId < 0 || m.seqId >= nContigs || (i && rec[i - 1].seqId > m.seqId)) atomicOr(&diag[1], 1ull);
  const long long len = (long long)m.wpos_end - m.wpos;
  if (len > 0) atomicMax(&diag[2], (unsigned long long)len);
}
// first record of every contig (records are grouped by ascending seqId) and the wpos of its last record
__global__ void k_contig_ranges(size_t n, const mm_minmer* __restrict__ rec, int nContigs, int64_t* __restrict__ first, int32_t* __restrict__ lastW) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > nContigs) return;
  size_t lo = 0, hi = n;
  while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (rec[mid].seqId < s) lo = mid + 1; else hi = mid; }
  first[s] = (int64_t)lo;
  if (s < nContigs) {
    size_t l2 = lo, h2 = n;
    while (l2 < h2) { const size_t mid = (l2 + h2) >> 1; if (rec[mid].seqId <= s) l2 = mid + 1; else h2 = mid; }
    lastW[s] = l2 > lo ? rec[l2 - 1].wpos : -1;
  }
}
// two events per record: insert at wpos, eviction at wpos_end; key = contig << 32 | pos*2 + isInsert, value = 2*record + isEviction
__global__ void __launch_bounds__(256)
k_event_keys(size_t n, const mm_minmer* __restrict__ rec, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const mm_minmer m = rec[i];
  key[2 * i] = ((uint64_t)(uint32_t)m.seqId << 32) | ((uint64_t)(uint32_t)m.wpos * 2ull + 1ull);
  key[2 * i + 1] = ((uint64_t)(uint32_t)m.seqId << 32) | ((uint64_t)(uint32_t)m.wpos_end * 2ull);
  val[2 * i] = (uint32_t)(2 * i); val[2 * i + 1] = (uint32_t)(2 * i + 1);
}
__global__ void __launch_bounds__(256)
k_event_fill(size_t nEv, const mm_minmer* __restrict__ rec, const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint32_t* __restrict__ evKey,
             uint32_t* __restrict__ evAux, uint64_t* __restrict__ evHash) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nEv) return;
  const mm_minmer m = rec[val[e] >> 1];
  evKey[e] = (uint32_t)key[e];
  evAux[e] = (val[e] & 1u) ? 0u : ((uint32_t)m.wpos_end | (m.strand < 0 ? 0x80000000u : 0u));
  evHash[e] = m.hash;
}
__global__ void k_contig_event_off(size_t nEv, const uint64_t* __restrict__ key, int nContigs, int64_t* __restrict__ off) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > nContigs) return;
  size_t lo = 0, hi = nEv; const uint64_t want = (uint64_t)(uint32_t)s << 32;
  while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (key[mid] < want) lo = mid + 1; else hi = mid; }
  off[s] = (int64_t)lo;
}
// per block of MM_OPEN_BLOCK positions: first event at or behind its start (evBlock), and -- one wave per block -- the records open at
// its start, wpos < B < wpos_end, in minmerIndex order: counted (WRITE = false), then written behind the scanned offsets
__device__ __forceinline__ int block_contig(int64_t blk, const int64_t* __restrict__ contigBlock, int nContigs) {
  int lo = 0, hi = nContigs;                                                   // last contig with contigBlock <= blk
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (contigBlock[mid] <= blk) lo = mid; else hi = mid; }
  return lo;
}
__global__ void __launch_bounds__(256)
k_block_events(int64_t nBlocks, const int64_t* __restrict__ contigBlock, int nContigs, const int64_t* __restrict__ contigOff, const uint32_t* __restrict__ evKey,
               int64_t nEv, int64_t* __restrict__ evBlock) {
  const int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blk > nBlocks) return;
  if (blk == nBlocks) { evBlock[blk] = nEv; return; }
  const int s = block_contig(blk, contigBlock, nContigs);
  const uint32_t want = (uint32_t)((blk - contigBlock[s]) << MM_OPEN_BLOCK_SHIFT) * 2u;
  int64_t lo = contigOff[s], hi = contigOff[s + 1];
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (evKey[mid] < want) lo = mid + 1; else hi = mid; }
  evBlock[blk] = lo;
}
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_block_open(int64_t nBlocks, const int64_t* __restrict__ contigBlock, int nContigs, const int64_t* __restrict__ contigRec, const mm_minmer* __restrict__ rec,
             int maxLen, int32_t* __restrict__ cnt, const int64_t* __restrict__ blockOff, uint32_t* __restrict__ opKey, uint32_t* __restrict__ opAux,
             uint64_t* __restrict__ opHash) {
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= nBlocks) return;
  const int lane = threadIdx.x & 63;
  const int s = block_contig(blk, contigBlock, nContigs);
  const int64_t B = (blk - contigBlock[s]) << MM_OPEN_BLOCK_SHIFT;
  const int64_t r0 = contigRec[s], r1 = contigRec[s + 1];
  auto lower = [&](int64_t pos) { int64_t lo = r0, hi = r1; while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)rec[mid].wpos < pos) lo = mid + 1; else hi = mid; } return lo; };
  const int64_t lo = lower(B - maxLen), hi = lower(B);                          // a record is at most maxLen long: candidates have wpos in [B - maxLen, B)
  int64_t at = WRITE ? blockOff[blk] : 0; int total = 0;
  for (int64_t base = lo; base < hi; base += 64) {
    const int64_t i = base + lane;
    mm_minmer m; bool open = false;
    if (i < hi) { m = rec[i]; open = (int64_t)m.wpos_end > B; }
    const uint64_t mask = __ballot(open);
    if (WRITE && open) {
      const int64_t o = at + __popcll(mask & ((1ull << lane) - 1ull));
      opKey[o] = (uint32_t)m.wpos * 2u + 1u; opAux[o] = (uint32_t)m.wpos_end | (m.strand < 0 ? 0x80000000u : 0u); opHash[o] = m.hash;
    }
    at += __popcll(mask); total += __popcll(mask);
  }
  if (!WRITE && lane == 0) cnt[blk] = total;
}
// seed table (linear probing, load <= 0.5) + presence bitmap
__global__ void __launch_bounds__(256)
k_ht_clear(size_t cap, HtSlot* __restrict__ ht) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < cap) { ht[i].key = MM_EMPTY; ht[i].val = 0; }
}
__global__ void __launch_bounds__(256)
k_ht_insert(size_t nk, const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keyOff, const uint8_t* __restrict__ freq, HtSlot* __restrict__ ht,
            uint64_t mask, unsigned long long* __restrict__ filter, uint64_t filterMask, uint8_t* __restrict__ tags /* non-null: bucketised placement + tag bytes */,
            unsigned long long* __restrict__ diag /* [3] |= 1 value overflow, |= 2 duplicate key */) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nk) return;
  const uint64_t key = keys[k];
  uint64_t off = keyOff[k], cnt = keyOff[k + 1] - keyOff[k];
  const bool f = freq[k] != 0;
  // a frequent seed is removed from the query sketch before any lookup (getSeedHits, computeMap.hpp:834-837): its point list is never
  // read on the device, so however long it is (satellite arrays) it needs no room in the packed value
  if (f) { off = 0; cnt = 0; }
  if (cnt >= (1ull << 23) || off >= (1ull << 40)) { atomicOr(&diag[3], 1ull); return; }
  const uint64_t val = (off << 24) | (cnt << 1) | (f ? 1ull : 0ull);
  if (tags) {
    // tagged table: the key goes into the first bucket, from its home bucket on, that still has a free slot (no deletions: the
    // buckets before it stay full, which is what ends an unsuccessful look-up at the first bucket with an empty tag)
    uint64_t b = (key & mask) & ~(uint64_t)(MM_TAG_BUCKET - 1);
    const uint32_t start = (uint32_t)(key >> 40);                      // spread the first attempts of a bucket's keys over its slots
    while (true) {
      for (uint32_t i = 0; i < MM_TAG_BUCKET; i++) {
        const uint64_t slot = b + ((start + i) & (MM_TAG_BUCKET - 1));
        const unsigned long long prev = atomicCAS((unsigned long long*)&ht[slot].key, (unsigned long long)MM_EMPTY, (unsigned long long)key);
        if (prev == MM_EMPTY) { ht[slot].val = val; tags[slot] = (uint8_t)mm_seed_tag(key); return; }
        if (prev == key) { atomicOr(&diag[3], 2ull); return; }
      }
      b = (b + MM_TAG_BUCKET) & mask;
    }
  }
  uint64_t slot = key & mask;
  while (true) {
    const unsigned long long prev = atomicCAS((unsigned long long*)&ht[slot].key, (unsigned long long)MM_EMPTY, (unsigned long long)key);
    if (prev == MM_EMPTY) { ht[slot].val = val; break; }
    if (prev == key) { atomicOr(&diag[3], 2ull); break; }
    slot = (slot + 1) & mask;
  }
  if (filterMask) atomicOr(&filter[mm_filter_word(key, filterMask)], (unsigned long long)mm_filter_bits(key));
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_order_keys(int c0, int n, const int32_t* __restrict__ key, int shift, uint32_t maxKey, uint32_t* __restrict__ kOut, int32_t* __restrict__ vOut) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t k = (uint32_t)key[c0 + i] >> shift; if (k > maxKey) k = maxKey;
  kOut[i] = maxKey - k;                                              // ascending sort of the complement = descending order
  vOut[i] = c0 + i;
}

namespace {

struct Tmp {                        // scratch device buffers of one build, released at the end
  std::vector<DevBuf*> all;
  DevBuf* make() { all.push_back(new DevBuf()); return all.back(); }
  ~Tmp() { for (DevBuf* b : all) { b->release(); delete b; } }
};

template <class K, class V>
int sort_pairs(mm_ctx* c, DevBuf& tmp, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned beginBit, unsigned endBit) {
  size_t bytes = 0;
  MM_HIP(c, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, beginBit, endBit, c->stream));
  MM_HIP(c, tmp.ensure(bytes + 256));
  MM_HIP(c, rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, beginBit, endBit, c->stream));
  return MM_OK;
}
template <class T>
int excl_scan(mm_ctx* c, DevBuf& tmp, const T* in, T* out, size_t n) {
  size_t bytes = 0;
  MM_HIP(c, rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), c->stream));
  MM_HIP(c, tmp.ensure(bytes + 256));
  MM_HIP(c, rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), c->stream));
  return MM_OK;
}
unsigned bits_for(uint64_t v) { unsigned b = 1; while (b < 64 && (v >> b)) b++; return b; }

}  // namespace

// dRec: minmerIndex (after dropFreqSeedSet) on the device; dKeys / dKeyOff / dKeyFreq / dPtKeys: the lookup map in ascending key order.
// Builds every array of DeviceIndex.  The key arrays and the packed points are adopted by the index (I.keys, I.keyOff, I.keyFreq, I.ptKeys).
int mm_flatten_device_index(mm_ctx* c, const mm_minmer* dRec, size_t n, size_t nk, size_t np, const int32_t* contigLen, const int32_t* refGroup, size_t nContigs) {
  DeviceIndex& I = c->idx;
  I.ready = false;
  Tmp T;
  DevBuf& scratch = *T.make();
  unsigned long long* diag = c->dCounters.as<unsigned long long>() + 20;       // [20..23]
  MM_HIP(c, hipMemsetAsync(diag, 0, 32, c->stream));
  const int nC = (int)nContigs;
  if (n) K_LAUNCH(k_check_records, n, n, dRec, nC, diag);
  DevBuf& dFirst = *T.make(); DevBuf& dLastW = *T.make();
  MM_HIP(c, dFirst.ensure((nContigs + 1) * 8)); MM_HIP(c, dLastW.ensure(nContigs * 4 + 4));
  hipLaunchKernelGGL(k_contig_ranges, dim3((unsigned)((nContigs + 1 + 255) / 256)), dim3(256), 0, c->stream, n, dRec, nC, dFirst.as<int64_t>(), dLastW.as<int32_t>());
  std::vector<int32_t> lastW(nContigs);
  unsigned long long hd[4];
  MM_HIP(c, hipMemcpyAsync(lastW.data(), dLastW.p, nContigs * 4, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipMemcpyAsync(hd, diag, 32, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  if (hd[1]) { c->err = "mm_index_upload: minmerIndex is not grouped by ascending seqId, or holds a negative position / unknown seqId"; return MM_ERR_ARG; }
  const int maxLen = (int)hd[2];
  // ---- event stream
  const size_t nEv = 2 * n;
  MM_HIP(c, I.evKey.ensure(nEv * 4 + 256)); MM_HIP(c, I.evAux.ensure(nEv * 4 + 256)); MM_HIP(c, I.evHash.ensure(nEv * 8 + 512));
  MM_HIP(c, I.contigOff.ensure((nContigs + 1) * 8));
  if (n) {
    DevBuf& k0 = *T.make(); DevBuf& k1 = *T.make(); DevBuf& v0 = *T.make(); DevBuf& v1 = *T.make();
    MM_HIP(c, k0.ensure(nEv * 8)); MM_HIP(c, k1.ensure(nEv * 8)); MM_HIP(c, v0.ensure(nEv * 4)); MM_HIP(c, v1.ensure(nEv * 4));
    K_LAUNCH(k_event_keys, n, n, dRec, k0.as<uint64_t>(), v0.as<uint32_t>());
    // stable: equal keys keep minmerIndex order -- inserts at one position in index order, evictions in the order a stable sort by wpos_end gives
    int rc = sort_pairs(c, scratch, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(), v1.as<uint32_t>(), nEv, 0u, 32u + bits_for(nContigs));
    if (rc != MM_OK) return rc;
    K_LAUNCH(k_event_fill, nEv, nEv, dRec, k1.as<uint64_t>(), v1.as<uint32_t>(), I.evKey.as<uint32_t>(), I.evAux.as<uint32_t>(), I.evHash.as<uint64_t>());
    hipLaunchKernelGGL(k_contig_event_off, dim3((unsigned)((nContigs + 1 + 255) / 256)), dim3(256), 0, c->stream, nEv, k1.as<uint64_t>(), nC, I.contigOff.as<int64_t>());
    MM_HIP(c, hipGetLastError());
    MM_HIP(c, hipStreamSynchronize(c->stream));
    k0.release(); k1.release(); v0.release(); v1.release();
  } else MM_HIP(c, hipMemsetAsync(I.contigOff.p, 0, (nContigs + 1) * 8, c->stream));
  // ---- blocks: first event of every block, records open at every block boundary
  std::vector<int64_t> contigBlock(nContigs + 1, 0);
  for (size_t s = 0; s < nContigs; s++) {
    const int64_t lastPos = std::max<int64_t>(contigLen[s], lastW[s]);
    contigBlock[s + 1] = contigBlock[s] + (lastPos >> MM_OPEN_BLOCK_SHIFT) + 1;
  }
  const int64_t nBlocks = contigBlock[nContigs];
  MM_HIP(c, I.contigBlock.ensure((nContigs + 1) * 8)); MM_HIP(c, I.evBlock.ensure((size_t)(nBlocks + 1) * 8)); MM_HIP(c, I.blockOff.ensure((size_t)(nBlocks + 1) * 8));
  MM_HIP(c, hipMemcpyAsync(I.contigBlock.p, contigBlock.data(), (nContigs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  K_LAUNCH(k_block_events, nBlocks + 1, nBlocks, I.contigBlock.as<int64_t>(), nC, I.contigOff.as<int64_t>(), I.evKey.as<uint32_t>(), (int64_t)nEv, I.evBlock.as<int64_t>());
  int64_t nOpen = 0;
  {
    DevBuf& cnt = *T.make();
    MM_HIP(c, cnt.ensure((size_t)nBlocks * 4 + 64));
    hipLaunchKernelGGL((k_block_open<false>), dim3((unsigned)((nBlocks + 3) / 4)), dim3(256), 0, c->stream, nBlocks, I.contigBlock.as<int64_t>(), nC, dFirst.as<int64_t>(),
                       dRec, maxLen, cnt.as<int32_t>(), (const int64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint64_t*)nullptr);
    MM_HIP(c, hipGetLastError());
    int rc = mm_scan_i32_to_i64(c, nBlocks, cnt.as<int32_t>(), I.blockOff.as<int64_t>(), &nOpen);
    if (rc != MM_OK) return rc;
    MM_HIP(c, hipMemcpyAsync(I.blockOff.as<int64_t>() + nBlocks, &nOpen, 8, hipMemcpyHostToDevice, c->stream));
    MM_HIP(c, I.opKey.ensure((size_t)nOpen * 4 + 256)); MM_HIP(c, I.opAux.ensure((size_t)nOpen * 4 + 256)); MM_HIP(c, I.opHash.ensure((size_t)nOpen * 8 + 512));
    hipLaunchKernelGGL((k_block_open<true>), dim3((unsigned)((nBlocks + 3) / 4)), dim3(256), 0, c->stream, nBlocks, I.contigBlock.as<int64_t>(), nC, dFirst.as<int64_t>(),
                       dRec, maxLen, (int32_t*)nullptr, I.blockOff.as<int64_t>(), I.opKey.as<uint32_t>(), I.opAux.as<uint32_t>(), I.opHash.as<uint64_t>());
    MM_HIP(c, hipGetLastError());
    MM_HIP(c, hipStreamSynchronize(c->stream));
  }
  // ---- seed table + presence bitmap
  size_t cap = 16; while (cap < 2 * nk + 2) cap <<= 1;
  // presence filter in front of the table (mm_filter_bits: 3 bits per key inside one 64-bit word): 4..8 bits per key, i.e. 2 MB for the
  // ~3 M keys of a 100 Mbp index -- resident in an XCD's L2, ~7 % false positives.  Most query seeds are absent from the index
  // (sequencing errors): they cost one cached 8-byte load instead of a table slot fetched over the fabric.  Beyond MM_FILTER_MAX_MIB
  // (default 64) the filter is off: against a 3 Gbp index no size of it paid (profiles/r02c_log_occupancy_filter.txt).
  uint64_t bitsPerKey = 4;
  if (const char* e = getenv("MM_FILTER_BITS_PER_KEY")) bitsPerKey = strtoull(e, nullptr, 10);
  uint64_t fbits = 4096; while (bitsPerKey && fbits < bitsPerKey * (uint64_t)nk) fbits <<= 1;
  if (!bitsPerKey) fbits = 0;
  uint64_t maxMiB = 64;
  if (const char* e = getenv("MM_FILTER_MAX_MIB")) maxMiB = strtoull(e, nullptr, 10);
  if (fbits / 8 > (maxMiB << 20)) fbits = 0;
  // Tag layer instead of the filter for tables beyond MM_SEED_TAGS_MIN_MIB (default 1024 MiB of slots: indexes of more than ~30 M keys,
  // where neither the table nor a filter of any useful size stays cached): one tag byte per slot, buckets of 16 slots.  MM_SEED_TAGS=1 / 0
  // forces it on / off (tests run the small parity cases both ways).
  bool tagged = cap * 16 > ((size_t)1024 << 20);
  if (const char* e = getenv("MM_SEED_TAGS_MIN_MIB")) tagged = cap * 16 > ((size_t)strtoull(e, nullptr, 10) << 20);
  if (const char* e = getenv("MM_SEED_TAGS")) tagged = atoi(e) != 0;
  if (cap < 4 * MM_TAG_BUCKET) tagged = false;
  if (tagged) fbits = 0;
  MM_HIP(c, I.htSlots.ensure(cap * 16)); MM_HIP(c, I.filter.ensure((fbits ? fbits / 8 : 4) + 64));
  MM_HIP(c, I.htTags.ensure(tagged ? cap + 64 : 64));
  K_LAUNCH(k_ht_clear, cap, cap, I.htSlots.as<HtSlot>());
  MM_HIP(c, hipMemsetAsync(I.filter.p, 0, (fbits ? fbits / 8 : 4), c->stream));
  if (tagged) MM_HIP(c, hipMemsetAsync(I.htTags.p, 0, cap + 64, c->stream));
  if (nk) K_LAUNCH(k_ht_insert, nk, nk, I.keys.as<uint64_t>(), I.keyOff.as<uint64_t>(), I.keyFreq.as<uint8_t>(), I.htSlots.as<HtSlot>(), (uint64_t)(cap - 1),
                   I.filter.as<unsigned long long>(), fbits ? fbits / 64 - 1 : 0ull, tagged ? I.htTags.as<uint8_t>() : (uint8_t*)nullptr, diag);
  I.tagged = tagged;
  MM_HIP(c, hipGetLastError());
  std::vector<int32_t> grp(nContigs, 0);
  if (refGroup) grp.assign(refGroup, refGroup + nContigs);
  MM_HIP(c, I.contigLen.ensure(nContigs * 4)); MM_HIP(c, I.refGroup.ensure(nContigs * 4));
  MM_HIP(c, hipMemcpyAsync(I.contigLen.p, contigLen, nContigs * 4, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemcpyAsync(I.refGroup.p, grp.data(), nContigs * 4, hipMemcpyHostToDevice, c->stream));
  MM_HIP(c, hipMemcpyAsync(hd, diag, 32, hipMemcpyDeviceToHost, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  if (hd[3] & 1ull) { c->err = "mm_index_upload: a non-frequent seed with 2^23 or more interval points (or 2^40 points in total) does not fit the packed table value"; return MM_ERR_ARG; }
  if (hd[3] & 2ull) { c->err = "mm_index_upload: duplicate key"; return MM_ERR_ARG; }
  I.filterMask = fbits ? fbits / 64 - 1 : 0;                                   // word mask
  I.nRec = n; I.nKeys = nk; I.nPoints = np; I.nContigs = nContigs; I.htCap = cap; I.nOpen = (size_t)nOpen; I.ready = true;
  return MM_OK;
}

// Sketch::index + the frequency filter on the device, from minmerIndex BEFORE the drop (host array, reference layout), then the flat
// device index.  Leaves nothing on the host but the frequent-seed list (small) and, with MM_OPT_KEEP_FULL_INDEX, the caller's records.
int mm_finalize_index_device(mm_ctx* c, const std::vector<std::pair<const mm_minmer*, size_t>>& parts, float kmerPctThreshold, const int32_t* contigLen,
                             const int32_t* refGroup, size_t nContigs) {
  DeviceIndex& I = c->idx;
  I.ready = false;
  size_t nAll = 0;
  for (const auto& p : parts) nAll += p.second;
  if (nAll >= (1ull << 31)) { c->err = "mm_index_build: more than 2^31 minmer records"; return MM_ERR_CAPACITY; }
  Tmp T;
  DevBuf& scratch = *T.make();
  DevBuf& dAll = *T.make();
  MM_HIP(c, c->dCounters.ensure(256));
  unsigned long long* diag = c->dCounters.as<unsigned long long>() + 16;       // [16] sort check, [17] nFreq, [18] list cursor, [19] threshold
  MM_HIP(c, hipMemsetAsync(diag, 0, 32, c->stream));
  MM_HIP(c, dAll.ensure(nAll * sizeof(mm_minmer) + 64));
  {
    size_t at = 0;                                     // the contigs' records go up one after the other: no concatenated host copy
    for (const auto& p : parts) {
      if (p.second) MM_HIP(c, hipMemcpyAsync(dAll.as<mm_minmer>() + at, p.first, p.second * sizeof(mm_minmer), hipMemcpyHostToDevice, c->stream));
      at += p.second;
    }
  }
  size_t nk = 0, nRuns = 0;
  DevBuf& sIdx = *T.make(); DevBuf& flags = *T.make(); DevBuf& pre = *T.make();
  if (nAll) {
    DevBuf& h0 = *T.make(); DevBuf& h1 = *T.make(); DevBuf& i0 = *T.make();
    MM_HIP(c, h0.ensure(nAll * 8)); MM_HIP(c, h1.ensure(nAll * 8)); MM_HIP(c, i0.ensure(nAll * 4)); MM_HIP(c, sIdx.ensure(nAll * 4));
    K_LAUNCH(k_idx_keys, nAll, nAll, dAll.as<mm_minmer>(), h0.as<uint64_t>(), i0.as<uint32_t>());
    int rc = sort_pairs(c, scratch, h0.as<uint64_t>(), h1.as<uint64_t>(), i0.as<uint32_t>(), sIdx.as<uint32_t>(), nAll, 0u, 64u);
    if (rc != MM_OK) return rc;
    MM_HIP(c, hipStreamSynchronize(c->stream));
    h0.release(); i0.release();
    MM_HIP(c, flags.ensure((nAll + 1) * 8)); MM_HIP(c, pre.ensure((nAll + 1) * 8));
    K_LAUNCH(k_idx_flags, nAll, nAll, dAll.as<mm_minmer>(), h1.as<uint64_t>(), sIdx.as<uint32_t>(), flags.as<uint64_t>(), diag);
    MM_HIP(c, hipMemsetAsync(flags.as<uint64_t>() + nAll, 0, 8, c->stream));
    rc = excl_scan(c, scratch, flags.as<uint64_t>(), pre.as<uint64_t>(), nAll + 1);      // entry nAll = the totals
    if (rc != MM_OK) return rc;
    uint64_t tot = 0;
    MM_HIP(c, hipMemcpyAsync(&tot, pre.as<uint64_t>() + nAll, 8, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
    nRuns = (size_t)(tot & 0xffffffffull); nk = (size_t)(tot >> 32);
    MM_HIP(c, I.keys.ensure(nk * 8 + 64)); MM_HIP(c, I.keyOff.ensure((nk + 1) * 8 + 64)); MM_HIP(c, I.keyFreq.ensure(nk + 64)); MM_HIP(c, I.ptKeys.ensure(2 * nRuns * 8 + 64));
    DevBuf& runSeq = *T.make();
    MM_HIP(c, runSeq.ensure(nRuns * 4 + 64));
    K_LAUNCH(k_idx_open, nAll, nAll, dAll.as<mm_minmer>(), h1.as<uint64_t>(), sIdx.as<uint32_t>(), flags.as<uint64_t>(), pre.as<uint64_t>(), I.keys.as<uint64_t>(),
             I.keyOff.as<uint64_t>(), I.ptKeys.as<uint64_t>(), runSeq.as<int32_t>());
    K_LAUNCH(k_idx_close, nAll, nAll, dAll.as<mm_minmer>(), sIdx.as<uint32_t>(), flags.as<uint64_t>(), pre.as<uint64_t>(), runSeq.as<int32_t>(), I.ptKeys.as<uint64_t>());
    const uint64_t endOff = 2 * (uint64_t)nRuns;
    MM_HIP(c, hipMemcpyAsync(I.keyOff.as<uint64_t>() + nk, &endOff, 8, hipMemcpyHostToDevice, c->stream));
    MM_HIP(c, hipGetLastError());
    MM_HIP(c, hipStreamSynchronize(c->stream));
    h1.release(); runSeq.release();
  } else {
    MM_HIP(c, I.keys.ensure(64)); MM_HIP(c, I.keyOff.ensure(64)); MM_HIP(c, I.keyFreq.ensure(64)); MM_HIP(c, I.ptKeys.ensure(64));
    MM_HIP(c, hipMemsetAsync(I.keyOff.p, 0, 8, c->stream));
  }
  // ---- frequency filter (winSketch.hpp:410-504)
  int32_t freqThreshold = 0x7fffffff;
  c->hFreq.clear();
  size_t nKeep = nAll;
  const mm_minmer* dRec = dAll.as<mm_minmer>();
  DevBuf& dMin = *T.make();
  if (nk) {
    DevBuf& c0 = *T.make(); DevBuf& c1 = *T.make();
    MM_HIP(c, c0.ensure(nk * 4)); MM_HIP(c, c1.ensure(nk * 4));
    K_LAUNCH(k_key_counts, nk, nk, I.keyOff.as<uint64_t>(), c0.as<uint32_t>());
    size_t bytes = 0;
    MM_HIP(c, rocprim::radix_sort_keys(nullptr, bytes, c0.as<uint32_t>(), c1.as<uint32_t>(), nk, 0u, 32u, c->stream));
    MM_HIP(c, scratch.ensure(bytes + 256));
    MM_HIP(c, rocprim::radix_sort_keys(scratch.p, bytes, c0.as<uint32_t>(), c1.as<uint32_t>(), nk, 0u, 32u, c->stream));
    const int64_t total = (int64_t)nk;
    const int64_t toIgnore = (int64_t)(total * kmerPctThreshold / 100);        // int64 * float / int, as winSketch.hpp:425
    int32_t* dThr = (int32_t*)(diag + 3);
    hipLaunchKernelGGL(k_freq_threshold, dim3(1), dim3(64), 0, c->stream, nk, c1.as<uint32_t>(), (unsigned long long)(toIgnore < 0 ? 0 : toIgnore), dThr);
    K_LAUNCH(k_key_freq, nk, nk, I.keyOff.as<uint64_t>(), dThr, I.keyFreq.as<uint8_t>(), diag + 1);
    unsigned long long hd[4];
    MM_HIP(c, hipMemcpyAsync(hd, diag, 32, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
    if (hd[0] & 1ull) { c->err = "mm_index_build: the device sort did not keep the records of a hash in minmerIndex order"; return MM_ERR_STATE; }
    std::memcpy(&freqThreshold, &hd[3], 4);
    const size_t nFreq = (size_t)hd[1];
    if (nFreq) {
      DevBuf& fl = *T.make();
      MM_HIP(c, fl.ensure(nFreq * 8));
      K_LAUNCH(k_freq_list, nk, nk, I.keys.as<uint64_t>(), I.keyFreq.as<uint8_t>(), fl.as<uint64_t>(), diag + 2);
      c->hFreq.resize(nFreq);
      MM_HIP(c, hipMemcpyAsync(c->hFreq.data(), fl.p, nFreq * 8, hipMemcpyDeviceToHost, c->stream));
      // dropFreqSeedSet: minmerIndex without the frequent hashes, order kept
      DevBuf& keep = *T.make(); DevBuf& pos = *T.make();
      MM_HIP(c, keep.ensure(nAll * 4 + 64)); MM_HIP(c, pos.ensure(nAll * 8 + 64));
      K_LAUNCH(k_keep_flags, nAll, nAll, sIdx.as<uint32_t>(), flags.as<uint64_t>(), pre.as<uint64_t>(), I.keyFreq.as<uint8_t>(), keep.as<int32_t>());
      MM_HIP(c, hipGetLastError());
      int64_t kept = 0;
      int rc = mm_scan_i32_to_i64(c, (int64_t)nAll, keep.as<int32_t>(), pos.as<int64_t>(), &kept);
      if (rc != MM_OK) return rc;
      nKeep = (size_t)kept;
      MM_HIP(c, dMin.ensure(nKeep * sizeof(mm_minmer) + 64));
      K_LAUNCH(k_compact_records, nAll, nAll, dAll.as<mm_minmer>(), keep.as<int32_t>(), pos.as<int64_t>(), dMin.as<mm_minmer>());
      MM_HIP(c, hipGetLastError());
      MM_HIP(c, hipStreamSynchronize(c->stream));
      std::sort(c->hFreq.begin(), c->hFreq.end());
      dRec = dMin.as<mm_minmer>();
      dAll.release(); keep.release(); pos.release();
    }
    c0.release(); c1.release();
  }
  sIdx.release(); flags.release(); pre.release();
  c->freqThreshold = freqThreshold;
  c->mapped = false;
  return mm_flatten_device_index(c, dRec, nKeep, nk, 2 * nRuns, contigLen, refGroup, nContigs);
}

// ---------------------------------------------------------------------------------------------
// host mirrors of a device-built index, on request (mm_index_download): minmerIndex from the insert events of the stream (they are
// in minmerIndex order), the lookup map from the key table + packed points
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_insert_flags(size_t nEv, const uint32_t* __restrict__ evKey, int32_t* __restrict__ f) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e < nEv) f[e] = (int32_t)(evKey[e] & 1u);
}
__global__ void __launch_bounds__(256)
k_records_from_events(size_t nEv, const uint32_t* __restrict__ evKey, const uint32_t* __restrict__ evAux, const uint64_t* __restrict__ evHash,
                      const int64_t* __restrict__ contigOff, int nContigs, const int64_t* __restrict__ pos, mm_minmer* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nEv || !(evKey[e] & 1u)) return;
  int lo = 0, hi = nContigs;                                                   // last contig with contigOff <= e
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (contigOff[mid] <= (int64_t)e) lo = mid; else hi = mid; }
  mm_minmer m; m.hash = evHash[e]; m.wpos = (int32_t)(evKey[e] >> 1); m.wpos_end = (int32_t)(evAux[e] & 0x7fffffffu); m.seqId = lo;
  m.strand = (evAux[e] >> 31) ? -1 : 1; m.pad_ = 0;
  out[pos[e]] = m;
}
__global__ void __launch_bounds__(256)
k_points_unpack(size_t np, size_t nk, const uint64_t* __restrict__ keys, const uint64_t* __restrict__ keyOff, const uint64_t* __restrict__ pt, mm_interval_point* __restrict__ out) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= np) return;
  size_t lo = 0, hi = nk;                                                      // last key with keyOff <= p
  while (hi - lo > 1) { const size_t mid = (lo + hi) >> 1; if (keyOff[mid] <= p) lo = mid; else hi = mid; }
  const uint64_t k = pt[p];
  mm_interval_point o; o.pos = (int32_t)(uint32_t)(k >> 1); o.pad0_ = 0; o.hash = keys[lo]; o.seqId = (int32_t)(k >> 33); o.side = (k & 1ull) ? 1 : -1;
  o.pad1_[0] = o.pad1_[1] = o.pad1_[2] = 0;
  out[p] = o;
}

int mm_mirror_minmers(mm_ctx* c) {
  DeviceIndex& I = c->idx;
  if (c->mirrorMinmers) return MM_OK;
  const size_t nEv = 2 * I.nRec;
  c->hMinmers.resize(I.nRec);
  if (I.nRec) {
    Tmp T;
    DevBuf& f = *T.make(); DevBuf& pos = *T.make(); DevBuf& out = *T.make();
    MM_HIP(c, f.ensure(nEv * 4)); MM_HIP(c, pos.ensure(nEv * 8)); MM_HIP(c, out.ensure(I.nRec * sizeof(mm_minmer)));
    K_LAUNCH(k_insert_flags, nEv, nEv, I.evKey.as<uint32_t>(), f.as<int32_t>());
    int64_t tot = 0;
    int rc = mm_scan_i32_to_i64(c, (int64_t)nEv, f.as<int32_t>(), pos.as<int64_t>(), &tot);
    if (rc != MM_OK) return rc;
    K_LAUNCH(k_records_from_events, nEv, nEv, I.evKey.as<uint32_t>(), I.evAux.as<uint32_t>(), I.evHash.as<uint64_t>(), I.contigOff.as<int64_t>(), (int)I.nContigs,
             pos.as<int64_t>(), out.as<mm_minmer>());
    MM_HIP(c, hipGetLastError());
    MM_HIP(c, hipMemcpyAsync(c->hMinmers.data(), out.p, I.nRec * sizeof(mm_minmer), hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
  }
  c->mirrorMinmers = true;
  return MM_OK;
}
int mm_mirror_map(mm_ctx* c) {
  DeviceIndex& I = c->idx;
  if (c->mirrorMap) return MM_OK;
  c->hKeys.resize(I.nKeys); c->hOffsets.assign(I.nKeys + 1, 0); c->hPoints.resize(I.nPoints);
  if (I.nKeys) {
    MM_HIP(c, hipMemcpyAsync(c->hKeys.data(), I.keys.p, I.nKeys * 8, hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipMemcpyAsync(c->hOffsets.data(), I.keyOff.p, (I.nKeys + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    Tmp T;
    DevBuf& out = *T.make();
    MM_HIP(c, out.ensure(I.nPoints * sizeof(mm_interval_point) + 64));
    if (I.nPoints) K_LAUNCH(k_points_unpack, I.nPoints, I.nPoints, I.nKeys, I.keys.as<uint64_t>(), I.keyOff.as<uint64_t>(), I.ptKeys.as<uint64_t>(), out.as<mm_interval_point>());
    MM_HIP(c, hipGetLastError());
    if (I.nPoints) MM_HIP(c, hipMemcpyAsync(c->hPoints.data(), out.p, I.nPoints * sizeof(mm_interval_point), hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
  }
  c->mirrorMap = true;
  return MM_OK;
}

// The indices c0 .. c0+n-1 ordered by descending key[i] >> shift (ties keep their order): the L2 sweep runs one candidate per lane, so
// a wave takes as long as its longest stream -- with the candidates taken in order of stream length the 64 of a wave are alike.
// ascending order of the n (key, value) pairs the caller has written to dL2Sort[0] (uint32 keys of `bits` bits) and dL2Sort[2] (int32 values)
int mm_order_pairs(mm_ctx* c, int n, unsigned bits, int32_t* dOrder) {
  if (n <= 0) return MM_OK;
  DevBuf& k0 = c->dL2Sort[0]; DevBuf& k1 = c->dL2Sort[1]; DevBuf& v0 = c->dL2Sort[2]; DevBuf& tmp = c->dL2Sort[3];
  MM_HIP(c, k1.ensure((size_t)n * 4 + 64));
  return sort_pairs(c, tmp, k0.as<uint32_t>(), k1.as<uint32_t>(), v0.as<int32_t>(), dOrder, (size_t)n, 0u, bits);
}
int mm_order_desc(mm_ctx* c, const int32_t* dKey, int c0, int n, int shift, int32_t* dOrder) {
  if (n <= 0) return MM_OK;
  const uint32_t maxKey = 0xFFFu;                                   // 12 bits of key: two radix passes
  DevBuf& k0 = c->dL2Sort[0]; DevBuf& k1 = c->dL2Sort[1]; DevBuf& v0 = c->dL2Sort[2]; DevBuf& tmp = c->dL2Sort[3];
  MM_HIP(c, k0.ensure((size_t)n * 4 + 64)); MM_HIP(c, k1.ensure((size_t)n * 4 + 64)); MM_HIP(c, v0.ensure((size_t)n * 4 + 64));
  hipLaunchKernelGGL(k_order_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c0, n, dKey, shift, maxKey, k0.as<uint32_t>(), v0.as<int32_t>());
  MM_HIP(c, hipGetLastError());
  return sort_pairs(c, tmp, k0.as<uint32_t>(), k1.as<uint32_t>(), v0.as<int32_t>(), dOrder, (size_t)n, 0u, 12u);
}

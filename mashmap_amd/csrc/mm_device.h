// mashmap_amd/csrc/mm_device.h -- device-side building blocks (gfx950, wave64).
//
// Bit-exact MurmurHash3_x64_128 (low word, seed 42) of the *ASCII* k-mer, both strands, computed
// from 2-bit packed bases.  Replaces CommonFunc::getHash (src/map/include/commonFunc.hpp:138) over
// MurmurHash3_x64_128 (src/common/murmur3.h:226) for the alphabet {A,C,G,T}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MM_SEED 42u
#define MM_C1 0x87c37b91114253d5ULL
#define MM_C2 0x4cf5ad432745937fULL
#define MM_HASH_MAX 0xFFFFFFFFFFFFFFFFULL

// 64-bit rotate left by a compile-time constant as two v_alignbit_b32 (the shift/or form costs 4-5 VALU instructions)
__device__ __forceinline__ uint64_t mm_rotl64(uint64_t x, int r) {
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  if (r == 32) return ((uint64_t)lo << 32) | hi;
  if (r < 32) return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 32 - r) << 32) | __builtin_amdgcn_alignbit(lo, hi, 32 - r);
  return ((uint64_t)__builtin_amdgcn_alignbit(lo, hi, 64 - r) << 32) | __builtin_amdgcn_alignbit(hi, lo, 64 - r);
}
__device__ __forceinline__ uint64_t mm_fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33; return k;
}
__device__ __forceinline__ uint64_t mm_mix_k1(uint64_t k) { k *= MM_C1; k = mm_rotl64(k, 31); k *= MM_C2; return k; }
__device__ __forceinline__ uint64_t mm_mix_k2(uint64_t k) { k *= MM_C2; k = mm_rotl64(k, 33); k *= MM_C1; return k; }

// 8 two-bit codes (A=0,C=1,G=2,T=3) in the low 16 bits -> nothing; 4 codes in the low 8 bits -> 4 ASCII bytes
__device__ __forceinline__ uint32_t mm_ascii4(uint32_t codes8) {
  uint32_t y = codes8 & 0xFFu;
  y = (y | (y << 12)) & 0x000F000Fu;
  y = (y | (y << 6)) & 0x03030303u;                 // one code per byte
  const uint32_t b0 = y & 0x01010101u, b1 = (y >> 1) & 0x01010101u, both = b0 & b1;
  // A=0x41, C=0x43 (+2), G=0x47 (+6), T=0x54 (+19 = 2+6+11)
  return 0x41414141u + (b0 << 1) + (b1 << 2) + (b1 << 1) + (both << 3) + (both << 1) + both;
}
// 16 codes -> 16 ASCII bytes (4 dwords)
__device__ __forceinline__ void mm_expand16(uint32_t w, uint32_t* out) {
  out[0] = mm_ascii4(w); out[1] = mm_ascii4(w >> 8); out[2] = mm_ascii4(w >> 16); out[3] = mm_ascii4(w >> 24);
}
// reverse the order of the sixteen 2-bit fields of w
__device__ __forceinline__ uint32_t mm_rev2(uint32_t w) {
  uint32_t r = __brev(w);
  return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}

// n (1..8) bytes starting at compile-time byte offset `off` of the dword stream A, zero-extended
template <int N>
__device__ __forceinline__ uint64_t mm_bytes(const uint32_t* A, int off) {
  const int w = off >> 2, sh = (off & 3) * 8;
  uint32_t lo, hi = 0;
  lo = sh ? __builtin_amdgcn_alignbit(A[w + 1], A[w], sh) : A[w];
  if (N > 4) hi = sh ? __builtin_amdgcn_alignbit(A[w + 2], A[w + 1], sh) : A[w + 1];
  if (N < 4) lo &= (1u << (8 * (N & 3))) - 1u;
  if (N > 4 && N < 8) hi &= (1u << (8 * (N & 3))) - 1u;
  return ((uint64_t)hi << 32) | lo;
}

// MurmurHash3_x64_128(key = K ASCII bytes at byte offset off of A, seed 42), low 64 bits.  Any K >= 1 (A must hold the bytes + one word of run-off).
template <int K>
__device__ __forceinline__ uint64_t mm_murmur_kmer(const uint32_t* A, int off) {
  uint64_t h1 = MM_SEED, h2 = MM_SEED;
  constexpr int NB = K / 16, TAIL = K & 15;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const uint64_t k1 = mm_bytes<8>(A, off + 16 * b), k2 = mm_bytes<8>(A, off + 16 * b + 8);
    h1 ^= mm_mix_k1(k1); h1 = mm_rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    h2 ^= mm_mix_k2(k2); h2 = mm_rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  if (TAIL > 8) h2 ^= mm_mix_k2(mm_bytes<(TAIL > 8 ? TAIL - 8 : 1)>(A, off + 16 * NB + 8));
  if (TAIL > 0) h1 ^= mm_mix_k1(mm_bytes<(TAIL > 8 ? 8 : (TAIL > 0 ? TAIL : 1))>(A, off + 16 * NB));
  h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
  h1 += h2; h2 += h1;
  h1 = mm_fmix64(h1); h2 = mm_fmix64(h2);
  return h1 + h2;
}

// The 16 k-mer positions of one strip: forward and reverse-complement ASCII streams of a 48-base window.
//   F[m]  = ascii(code[m])            m = 0..47   forward k-mer j starts at byte j
//   RC[m] = ascii(3 - code[47 - m])               reverse-complement of k-mer j starts at byte 48-K-j
struct MMTables;
struct MMStrip {
  uint32_t F[13], R[13];
  __device__ __forceinline__ void load(uint32_t w0, uint32_t w1, uint32_t w2) {
    mm_expand16(w0, F); mm_expand16(w1, F + 4); mm_expand16(w2, F + 8); F[12] = 0;
    mm_expand16(mm_rev2(~w2), R); mm_expand16(mm_rev2(~w1), R + 4); mm_expand16(mm_rev2(~w0), R + 8); R[12] = 0;
  }
  // same streams from the 256-entry LDS table: code byte q of the window gives F[q] and R[11-q] with one 8-byte read
  __device__ __forceinline__ void load(uint32_t w0, uint32_t w1, uint32_t w2, const MMTables& T);
};

__device__ __forceinline__ uint32_t mm_lane() { return threadIdx.x & 63u; }
// Lane masks straight from the compare (llvm.amdgcn.icmp: one v_cmp_*_e64 into an SGPR pair; inactive lanes read 0) and a 64-bit select on
// such a mask as two VOP3 v_cndmask.  cmp + two selects issue in 12.3 cycles this way against 16.3 in the vcc / VOP2 form the compiler picks
// for `a < b ? a : b` (profiles/r02_valu_rate.txt), and a test that is only ever used as a mask never becomes a 0/1 register first.
#define MM_ICMP_NE 33
#define MM_ICMP_ULT 36
__device__ __forceinline__ uint64_t mm_mask_lt64(uint64_t a, uint64_t b) { return __builtin_amdgcn_uicmpl(a, b, MM_ICMP_ULT); }
__device__ __forceinline__ uint64_t mm_mask_ne64(uint64_t a, uint64_t b) { return __builtin_amdgcn_uicmpl(a, b, MM_ICMP_NE); }
__device__ __forceinline__ uint64_t mm_mask_nz32(uint32_t a) { return __builtin_amdgcn_uicmp(a, 0u, MM_ICMP_NE); }
__device__ __forceinline__ uint64_t mm_mask_select64(uint64_t mask, uint64_t a, uint64_t b) {    // this lane's bit of mask ? a : b
  uint32_t lo, hi;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((uint32_t)b), "v"((uint32_t)a), "s"(mask));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((uint32_t)(b >> 32)), "v"((uint32_t)(a >> 32)), "s"(mask));
  return ((uint64_t)hi << 32) | lo;
}
// the wave's ballot of a predicate.  (__ballot() of the HIP headers takes an int: a bool that already lives in a lane mask is first
// turned into 0 / 1 and compared again -- two VALU instructions per ballot that this form does not have)
__device__ __forceinline__ uint64_t mm_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t mm_popc_below(uint64_t mask) {    // set bits of mask strictly below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Wave reductions / scans on the VALU's DPP lanes (no LDS round trip per step as with ds_bpermute shuffles).  All 64 lanes
// must be active.  quad_perm / row_half_mirror / row_mirror leave every lane with the result of its row of 16; the four
// rows are combined through v_readlane, so the result is wave-uniform (an SGPR).
template <int CTRL> __device__ __forceinline__ int mm_dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
#define MM_DPP_QUAD_1032 0xB1
#define MM_DPP_QUAD_2301 0x4E
#define MM_DPP_ROW_HALF_MIRROR 0x141
#define MM_DPP_ROW_MIRROR 0x140
#define MM_DPP_ROW_SHR(n) (0x110 + (n))
__device__ __forceinline__ int mm_wave_sum(int v) {
  v += mm_dpp0<MM_DPP_QUAD_1032>(v); v += mm_dpp0<MM_DPP_QUAD_2301>(v);
  v += mm_dpp0<MM_DPP_ROW_HALF_MIRROR>(v); v += mm_dpp0<MM_DPP_ROW_MIRROR>(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int mm_wave_max(int v) {              // values >= 0 (the fill of a masked DPP lane is 0)
  auto mx = [](int a, int b) { return a > b ? a : b; };
  v = mx(v, mm_dpp0<MM_DPP_QUAD_1032>(v)); v = mx(v, mm_dpp0<MM_DPP_QUAD_2301>(v));
  v = mx(v, mm_dpp0<MM_DPP_ROW_HALF_MIRROR>(v)); v = mx(v, mm_dpp0<MM_DPP_ROW_MIRROR>(v));
  return mx(mx(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), mx(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int mm_wave_excl_scan(int v) {      // exclusive prefix sum across the 64 lanes
  int x = v;                                                     // inclusive scan inside each row of 16 (row_shr fills with 0)
  x += mm_dpp0<MM_DPP_ROW_SHR(1)>(x); x += mm_dpp0<MM_DPP_ROW_SHR(2)>(x); x += mm_dpp0<MM_DPP_ROW_SHR(4)>(x); x += mm_dpp0<MM_DPP_ROW_SHR(8)>(x);
  const int r0 = __builtin_amdgcn_readlane(x, 15), r1 = __builtin_amdgcn_readlane(x, 31), r2 = __builtin_amdgcn_readlane(x, 47);
  const int row = (int)mm_lane() >> 4;
  const int off = row == 0 ? 0 : row == 1 ? r0 : row == 2 ? r0 + r1 : r0 + r1 + r2;
  return x + off - v;
}

// Presence filter in front of the seed table: 64-bit words, three bits per key inside ONE word (a word-blocked Bloom filter: one
// 8-byte load per query seed).  Word = bits 32.. of the hash, the three bit positions from its low 18 bits.  Sized so that it stays
// resident in an XCD's 4 MB L2 for indexes of a few hundred Mbp.
__host__ __device__ __forceinline__ uint64_t mm_filter_bits(uint64_t h) {
  return (1ull << (h & 63)) | (1ull << ((h >> 6) & 63)) | (1ull << ((h >> 12) & 63));
}
__host__ __device__ __forceinline__ uint64_t mm_filter_word(uint64_t h, uint64_t wordMask) { return (h >> 32) & wordMask; }

// Tag layer of the seed table (large indexes): buckets of 16 slots, one tag byte per slot, 0 = empty slot.  The bucket comes from the low
// bits of the hash (as the slot of the untagged table does), the tag from bits 29..36 (the minmer hashes of an index are the small ones
// of their windows: the top bits are zero).
#define MM_TAG_BUCKET 16
__host__ __device__ __forceinline__ uint32_t mm_seed_tag(uint64_t h) { const uint32_t t = (uint32_t)(h >> 29) & 0xFFu; return t ? t : 0xA7u; }

struct HtSlot { uint64_t key, val; };                          // one 16-byte slot: a probe costs one memory sector
#define MM_HT_EMPTY 0xFFFFFFFFFFFFFFFFULL
// tag bytes of one bucket of the tagged seed table (mm_internal.h: htTags) against the tag of a query seed: cand16 = slots whose tag
// equals it (a set bit above a matching or empty byte of the same 4-byte word may be spurious -- candidates are verified against the
// slot's key anyway), hasEmpty = the bucket still has a free slot, i.e. no key of this bucket lives further on
__device__ __forceinline__ void mm_tag_scan(uint4 t, uint32_t tag, uint32_t& cand16, bool& hasEmpty) {
  const uint32_t rep = tag * 0x01010101u;
  const uint32_t w[4] = {t.x, t.y, t.z, t.w};
  uint32_t c = 0, e = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t x = w[i] ^ rep;
    const uint32_t zc = (x - 0x01010101u) & ~x & 0x80808080u;          // zero bytes of x: bit 7 of the byte
    c |= ((((zc >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * i);          // the four flags side by side
    e |= (w[i] - 0x01010101u) & ~w[i] & 0x80808080u;
  }
  cand16 = c; hasEmpty = e != 0;
}

// the seed table as a kernel argument (k_lookup_l1, k_gather_points: mm_map.hip)
struct SeedTable { const HtSlot* ht; uint64_t mask; const uint64_t* filter; uint64_t filterMask; const uint8_t* tags; };

// ---------------------------------------------------------------------------------------------
// Strip hasher.  For K = 17..19 (one 16-byte block + a tail of K-16 <= 3 bytes; K = 19 is MashMap's default) the tail has
// at most 64 values, so its complete mix (k1*C1, rotl 31, *C2) comes from a 64-entry LDS table indexed by the 2-bit codes:
// 2 of the 10 64-bit multiplies per hash and the tail's byte assembly go away.  Bit-exact with mm_murmur_kmer<K>.
// (Also tried: the two block products k1*C1, k2*C2 by the sliding recurrence K1(j)*C1 = b(j)*C1 + ((K1(j+1)*C1) << 8) with
// 4-entry tables -- 4 fewer multiplies per hash on paper, but the same VALU instruction count and its LDS look-ups land on
// the dependency chain: 135-193 ms instead of 73 ms per 10 Gbp on MI355X.  The kernel issues VALU ~93 % of the time, so only
// fewer instructions help; see DESIGN.md.)
// ---------------------------------------------------------------------------------------------
struct MMTables {            // lives in LDS; filled by mm_tables_init
  uint2 ascii4[256];         // .x = ASCII of the 4 bases of a code byte; .y = ASCII of their reverse complement (both streams of MMStrip)
  uint64_t tailF[64];        // mix_k1 of the forward tail, indexed by the 2-bit codes of bases j+16.. (first base in the low bits)
  uint64_t tailR[64];        // mix_k1 of the reverse-complement tail, indexed by the codes of bases j..j+TAIL-1
};

__device__ __forceinline__ uint32_t mm_ascii1(uint32_t code) { return code == 0 ? 0x41u : code == 1 ? 0x43u : code == 2 ? 0x47u : 0x54u; }

template <int K> struct MMFastK { static constexpr bool value = (K >= 17 && K <= 19); };

template <int K>
__device__ __forceinline__ void mm_tables_init(MMTables& T, int tid, int nthr) {
  constexpr int TAIL = MMFastK<K>::value ? K - 16 : 0;
  for (int c = tid; c < 256; c += nthr) {
    uint32_t f = 0, r = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      f |= mm_ascii1((c >> (2 * i)) & 3) << (8 * i);
      r |= mm_ascii1(3 - ((c >> (2 * (3 - i))) & 3)) << (8 * i);
    }
    T.ascii4[c] = make_uint2(f, r);
  }
  for (int c = tid; c < 64; c += nthr) {
    uint64_t kf = 0, kr = 0;
#pragma unroll
    for (int i = 0; i < TAIL; i++) {
      kf |= (uint64_t)mm_ascii1((c >> (2 * i)) & 3) << (8 * i);                        // byte 16+i = base j+16+i
      kr |= (uint64_t)mm_ascii1(3 - ((c >> (2 * (TAIL - 1 - i))) & 3)) << (8 * i);     // byte 16+i = comp(base j+TAIL-1-i)
    }
    T.tailF[c] = TAIL ? mm_mix_k1(kf) : 0ull;
    T.tailR[c] = TAIL ? mm_mix_k1(kr) : 0ull;
  }
}

// 2-bit code of base i (0..47) of the 48-base window w[0..2]
#define MM_CODE(w, i) (((w)[(i) >> 4] >> (2 * ((i) & 15))) & 3u)

// block part from the ASCII streams (as mm_murmur_kmer), tail mix supplied by the caller (TAIL <= 8 bytes: no k2 tail)
template <int K>
__device__ __forceinline__ uint64_t mm_murmur_kmer_tail(const uint32_t* A, int off, uint64_t tailMix) {
  uint64_t h1 = MM_SEED, h2 = MM_SEED;
  constexpr int NB = K / 16;
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const uint64_t k1 = mm_bytes<8>(A, off + 16 * b), k2 = mm_bytes<8>(A, off + 16 * b + 8);
    h1 ^= mm_mix_k1(k1); h1 = mm_rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    h2 ^= mm_mix_k2(k2); h2 = mm_rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  h1 ^= tailMix;
  h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
  h1 += h2; h2 += h1;
  h1 = mm_fmix64(h1); h2 = mm_fmix64(h2);
  return h1 + h2;
}

// Hashes of the 16 k-mer positions of a strip on both strands; calls use(j, fwd, rc) for j = 0..15 in ascending order.
template <int K, int SL = 16, class Use>
__device__ __forceinline__ void mm_strip_hashes(uint32_t w0, uint32_t w1, uint32_t w2, const MMTables& T, Use&& use) {
  static_assert(SL == 16, "the ASCII-stream hasher works on strips of 16 positions");
  MMStrip st;
  st.load(w0, w1, w2, T);
  if constexpr (MMFastK<K>::value) {
    constexpr int TAIL = K - 16;
    const uint32_t w[3] = {w0, w1, w2};
#pragma unroll
    for (int j = 0; j < 16; j++) {
      uint32_t tf = 0, tr = 0;
#pragma unroll
      for (int i = 0; i < TAIL; i++) { tf |= MM_CODE(w, j + 16 + i) << (2 * i); tr |= MM_CODE(w, j + i) << (2 * i); }
      use(j, mm_murmur_kmer_tail<K>(st.F, j, T.tailF[tf]), mm_murmur_kmer_tail<K>(st.R, 48 - K - j, T.tailR[tr]));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; j++) use(j, mm_murmur_kmer<K>(st.F, j), mm_murmur_kmer<K>(st.R, 48 - K - j));
  }
}

// k-mers of 33..64 bases (the reference hashes any length, commonFunc.hpp:138): the 16 positions of a strip need 16 + K - 1 <= 79 bases,
// i.e. NW = 4 packed words for K <= 48, 5 beyond; the two ASCII streams of the window are expanded as in MMStrip and every position is
// hashed by the plain block-by-block MurmurHash3 (mm_murmur_kmer<K>: two to four 16-byte blocks + tail).  Not tuned: these sizes are
// off the reference's beaten path (k = 19), the point is that they run and are bit-exact.
template <int K> struct MMWideK { static constexpr bool value = (K > 32); static constexpr int NW = K <= 48 ? 4 : 5; };
template <int K, class Use>
__device__ __forceinline__ void mm_strip_hashes_wide(const uint32_t (&w)[MMWideK<K>::NW], const MMTables& T, Use&& use) {
  constexpr int NW = MMWideK<K>::NW, NQ = 4 * NW, W = 16 * NW;
  uint32_t F[NQ + 1], R[NQ + 1];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const uint2 e = T.ascii4[(w[q >> 2] >> (8 * (q & 3))) & 0xFFu];
    F[q] = e.x; R[NQ - 1 - q] = e.y;
  }
  F[NQ] = 0; R[NQ] = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) use(j, mm_murmur_kmer<K>(F, j), mm_murmur_kmer<K>(R, W - K - j));
}
// bit j (j = 0..15) of the result: any of the mask bits j .. j+K-1 of the 128-bit value hi:lo is set (the N test of a wide strip)
template <int K>
__device__ __forceinline__ uint32_t mm_window_or_wide(uint64_t lo, uint64_t hi) {
  const uint64_t km = K >= 64 ? ~0ull : ((1ull << (K & 63)) - 1ull);
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint64_t v = j ? ((lo >> j) | (hi << (64 - j))) : lo;
    r |= ((v & km) != 0ull ? 1u : 0u) << j;
  }
  return r;
}

__device__ __forceinline__ void MMStrip::load(uint32_t w0, uint32_t w1, uint32_t w2, const MMTables& T) {
  const uint32_t w[3] = {w0, w1, w2};
#pragma unroll
  for (int q = 0; q < 12; q++) {
    const uint2 e = T.ascii4[(w[q >> 2] >> (8 * (q & 3))) & 0xFFu];
    F[q] = e.x; R[11 - q] = e.y;
  }
  F[12] = 0; R[12] = 0;
}

// ---------------------------------------------------------------------------------------------
// Product-table strip hasher (K = 17..19).  The two block products of MurmurHash3 are linear in the key bytes:
//   k1 * C1 = ascii4(bases j..j+3) * C1 + ((ascii4(bases j+4..j+7) * C1) << 32)      (mod 2^64)
// and a group of 4 bases has only 256 values, so a 256-entry LDS table of 64-bit products per (strand, constant) replaces
// the ASCII reconstruction and 4 of the 8 remaining 64-bit multiplies per position pair: one 8-byte read, one 4-byte read
// (the low word of the entry of the next group) and one 32-bit add per product.  The look-ups depend only on the packed
// words, not on the hash chain, so they are issued ahead of it.  The reverse strand reads the same 8-bit windows through
// tables built from the reverse complement of the group.  Bit-exact with mm_murmur_kmer<K>.
// ---------------------------------------------------------------------------------------------
struct MMProdTables {        // lives in LDS (first thing in the dynamic segment); 9 KB
  uint64_t pf1[256];         // ascii4(c) * C1            forward strand, block word k1
  uint64_t pf2[256];         // ascii4(c) * C2            forward strand, block word k2
  uint64_t pr1[256];         // ascii4(revcomp c) * C1    reverse strand, k1
  uint64_t pr2[256];         // ascii4(revcomp c) * C2    reverse strand, k2
  uint64_t tailF[64];        // K = 17..19: complete mix of the tail ^ K; otherwise: products of the one group shorter than 4 bases
  uint64_t tailR[64];
};

// Word layout of MurmurHash3_x64_128 over K bytes, 16 <= K <= 32: K/16 blocks (k1, k2 of 8 bytes each), then tail words of
// T1 = min(K%16, 8) bytes (mixed like k1) and T2 = K%16 - T1 bytes (mixed like k2).  Every word is a sum of 4-byte groups; at most
// one group of the key is shorter than 4 bytes (PM bytes, in the k1 tail unless that tail is exactly 8 bytes).
template <int K> struct MMKeyLayout {
  static constexpr int NB = K / 16, T = K % 16, T1 = T > 8 ? 8 : T, T2 = T > 8 ? T - 8 : 0;
  static constexpr int PM = (T1 % 4) ? (T1 % 4) : (T2 % 4);
  static constexpr bool PART_IN_K2 = (T1 % 4) == 0 && (T2 % 4) != 0;
};

template <int K>
__device__ __forceinline__ void mm_tables_init(MMProdTables& T, int tid, int nthr) {
  for (int c = tid; c < 256; c += nthr) {
    uint32_t f = 0, r = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      f |= mm_ascii1((c >> (2 * i)) & 3) << (8 * i);
      r |= mm_ascii1(3 - ((c >> (2 * (3 - i))) & 3)) << (8 * i);
    }
    T.pf1[c] = (uint64_t)f * MM_C1; T.pf2[c] = (uint64_t)f * MM_C2;
    T.pr1[c] = (uint64_t)r * MM_C1; T.pr2[c] = (uint64_t)r * MM_C2;
  }
  if constexpr (MMFastK<K>::value) {
    // K = 17..19: the tail is the short group alone, so its complete mix (with the key length folded in) comes from the table
    constexpr int TAIL = K - 16;
    for (int c = tid; c < 64; c += nthr) {
      uint64_t kf = 0, kr = 0;
#pragma unroll
      for (int i = 0; i < TAIL; i++) {
        kf |= (uint64_t)mm_ascii1((c >> (2 * i)) & 3) << (8 * i);
        kr |= (uint64_t)mm_ascii1(3 - ((c >> (2 * (TAIL - 1 - i))) & 3)) << (8 * i);
      }
      T.tailF[c] = mm_mix_k1(kf) ^ (uint64_t)K;       // h1 ^= tail mix; h1 ^= len  in one xor
      T.tailR[c] = mm_mix_k1(kr) ^ (uint64_t)K;
    }
  } else {
    // otherwise: the products of the short group (PM bases) with the constant of the word it sits in
    constexpr int PM = MMKeyLayout<K>::PM;
    constexpr uint64_t CP = MMKeyLayout<K>::PART_IN_K2 ? MM_C2 : MM_C1;
    for (int c = tid; c < 64; c += nthr) {
      uint64_t kf = 0, kr = 0;
#pragma unroll
      for (int i = 0; i < PM; i++) {
        kf |= (uint64_t)mm_ascii1((c >> (2 * i)) & 3) << (8 * i);
        kr |= (uint64_t)mm_ascii1(3 - ((c >> (2 * (PM - 1 - i))) & 3)) << (8 * i);
      }
      T.tailF[c] = kf * CP;
      T.tailR[c] = kr * CP;
    }
  }
}

// byte offset (index * 8) into a table of 8-byte entries of the NB-base window that starts at base p (compile-time) of w[0..2]
template <int NB>
__device__ __forceinline__ uint32_t mm_win_off8(const uint32_t* w, int p) {
  const int word = p >> 4, bit = 2 * (p & 15);
  constexpr uint32_t mask = ((1u << (2 * NB)) - 1u) << 3;
  uint32_t x;
  if (bit + 2 * NB <= 32) x = bit >= 3 ? (w[word] >> (bit - 3)) : (w[word] << (3 - bit));
  else x = __builtin_amdgcn_alignbit(w[word + 1], w[word], bit - 3);
  return x & mask;
}
__device__ __forceinline__ uint64_t mm_lds64(const uint64_t* tab, uint32_t byteOff) { return *(const uint64_t*)((const unsigned char*)tab + byteOff); }
__device__ __forceinline__ uint32_t mm_lds32(const uint64_t* tab, uint32_t byteOff) { return *(const uint32_t*)((const unsigned char*)tab + byteOff); }

// h * 5 as one v_lshl_add_u64 ((h << 2) + h); hipcc's own choice is two v_mad_u64_u32 plus two moves to pair their operands
__device__ __forceinline__ uint64_t mm_times5(uint64_t h) {
  uint64_t r;
  asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(r) : "v"(h));
  return r;
}

// the hash from the two block products p1 = k1*C1, p2 = k2*C2 and the complete tail mix with the key length folded in (tailMixK = mix ^ K)
template <int K>
__device__ __forceinline__ uint64_t mm_murmur_from_products(uint64_t p1, uint64_t p2, uint64_t tailMixK) {
  uint64_t h1 = MM_SEED, h2 = MM_SEED;
  // (rotl(h1, 27) + seed) * 5 + c  =  rotl(h1, 27) * 5 + (5 * seed + c): one 64-bit add less
  h1 ^= mm_rotl64(p1, 31) * MM_C2; h1 = mm_rotl64(h1, 27); h1 = mm_times5(h1) + (0x52dce729ull + 5ull * MM_SEED);
  h2 ^= mm_rotl64(p2, 33) * MM_C1; h2 = mm_rotl64(h2, 31); h2 += h1; h2 = mm_times5(h2) + 0x38495ab5;
  h1 ^= tailMixK;
  h2 ^= (uint64_t)K;
  h1 += h2; h2 += h1;
  h1 = mm_fmix64(h1); h2 = mm_fmix64(h2);
  return h1 + h2;
}

// product of the NBYTES-byte key word at byte offset BYTEOFF of k-mer j with C1 (CI == 1) or C2, forward or reverse-complement
// strand: the 4-byte group at byte b of the forward key is the window of bases j+b.., of the reverse key (byte i = complement of
// base j+K-1-i) the window that ends at base j+K-1-b.
template <int K, bool RC, int CI, int BYTEOFF, int NBYTES>
__device__ __forceinline__ uint64_t mm_word_product(const uint32_t* w, int j, const MMProdTables& T) {
  constexpr int M0 = NBYTES < 4 ? NBYTES : 4, M1 = NBYTES - M0;
  const uint64_t* full = CI == 1 ? (RC ? T.pr1 : T.pf1) : (RC ? T.pr2 : T.pf2);
  const uint64_t* part = RC ? T.tailR : T.tailF;
  const int p0 = RC ? j + K - BYTEOFF - M0 : j + BYTEOFF;
  uint64_t P = mm_lds64(M0 == 4 ? full : part, mm_win_off8<M0>(w, p0));
  if constexpr (M1 > 0) {
    const int p1 = RC ? j + K - (BYTEOFF + 4) - M1 : j + BYTEOFF + 4;
    P += (uint64_t)mm_lds32(M1 == 4 ? full : part, mm_win_off8<M1>(w, p1)) << 32;
  }
  return P;
}

// the hash of k-mer j on one strand, every first-level product from the tables (16 <= K <= 32, K not in 17..19)
template <int K, bool RC>
__device__ __forceinline__ uint64_t mm_murmur_prod_general(const uint32_t* w, int j, const MMProdTables& T) {
  using L = MMKeyLayout<K>;
  uint64_t h1, h2;
  {
    const uint64_t p1 = mm_word_product<K, RC, 1, 0, 8>(w, j, T), p2 = mm_word_product<K, RC, 2, 8, 8>(w, j, T);
    h1 = MM_SEED ^ (mm_rotl64(p1, 31) * MM_C2); h1 = mm_rotl64(h1, 27); h1 = mm_times5(h1) + (0x52dce729ull + 5ull * MM_SEED);
    h2 = MM_SEED ^ (mm_rotl64(p2, 33) * MM_C1); h2 = mm_rotl64(h2, 31); h2 += h1; h2 = mm_times5(h2) + 0x38495ab5;
  }
  if constexpr (L::NB == 2) {
    const uint64_t p1 = mm_word_product<K, RC, 1, 16, 8>(w, j, T), p2 = mm_word_product<K, RC, 2, 24, 8>(w, j, T);
    h1 ^= mm_rotl64(p1, 31) * MM_C2; h1 = mm_rotl64(h1, 27); h1 += h2; h1 = mm_times5(h1) + 0x52dce729;
    h2 ^= mm_rotl64(p2, 33) * MM_C1; h2 = mm_rotl64(h2, 31); h2 += h1; h2 = mm_times5(h2) + 0x38495ab5;
  }
  if constexpr (L::T2 > 0) h2 ^= mm_rotl64(mm_word_product<K, RC, 2, 16 * L::NB + 8, (L::T2 > 0 ? L::T2 : 1)>(w, j, T), 33) * MM_C1;
  if constexpr (L::T1 > 0) h1 ^= mm_rotl64(mm_word_product<K, RC, 1, 16 * L::NB, (L::T1 > 0 ? L::T1 : 1)>(w, j, T), 31) * MM_C2;
  h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
  h1 += h2; h2 += h1;
  h1 = mm_fmix64(h1); h2 = mm_fmix64(h2);
  return h1 + h2;
}

// SL k-mer positions per strip (16, or more while the last k-mer still ends inside the 48-base window: SL + K - 1 <= 48)
template <int K, int SL = 16, class Use>
__device__ __forceinline__ void mm_strip_hashes(uint32_t w0, uint32_t w1, uint32_t w2, const MMProdTables& T, Use&& use) {
  static_assert(K >= 16 && K <= 32, "product tables need at least one 16-byte block");
  static_assert(SL >= 1 && SL + K - 1 <= 48, "a strip's k-mers must fit the 48-base window");
  const uint32_t w[3] = {w0, w1, w2};
  if constexpr (MMFastK<K>::value) {
    constexpr int TAIL = K - 16;
    uint32_t a[SL + K - 4];                         // a[p]: table offset of the 4-base group starting at base p (unused ones fold away)
#pragma unroll
    for (int p = 0; p < SL + K - 4; p++) a[p] = mm_win_off8<4>(w, p);
#pragma unroll
    for (int j = 0; j < SL; j++) {
      uint64_t f1 = mm_lds64(T.pf1, a[j]), f2 = mm_lds64(T.pf2, a[j + 8]);
      f1 += (uint64_t)mm_lds32(T.pf1, a[j + 4]) << 32; f2 += (uint64_t)mm_lds32(T.pf2, a[j + 12]) << 32;
      // reverse complement of k-mer j: byte i = comp(base j+K-1-i), so its 4-byte groups are the windows at j+K-4, j+K-8, ...
      uint64_t r1 = mm_lds64(T.pr1, a[j + K - 4]), r2 = mm_lds64(T.pr2, a[j + K - 12]);
      r1 += (uint64_t)mm_lds32(T.pr1, a[j + K - 8]) << 32; r2 += (uint64_t)mm_lds32(T.pr2, a[j + K - 16]) << 32;
      const uint64_t tf = mm_lds64(T.tailF, mm_win_off8<TAIL>(w, j + 16)), tr = mm_lds64(T.tailR, mm_win_off8<TAIL>(w, j));
      use(j, mm_murmur_from_products<K>(f1, f2, tf), mm_murmur_from_products<K>(r1, r2, tr));
    }
  } else {
#pragma unroll
    for (int j = 0; j < SL; j++) use(j, mm_murmur_prod_general<K, false>(w, j, T), mm_murmur_prod_general<K, true>(w, j, T));
  }
}

// table type of the strip hasher for a given K
template <int K, bool FAST = (K >= 16 && K <= 32)> struct MMTabsFor { using type = MMTables; };
template <int K> struct MMTabsFor<K, true> { using type = MMProdTables; };

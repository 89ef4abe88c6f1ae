// mashmap_amd/host/pack2bit.hpp -- makeUpperCaseAndValidDNA (src/map/include/commonFunc.hpp:97-107) + 2-bit packing on the HOST.
//
// The device layout of a batch of reads (`dBases2`, `dNmask`) is 2 bit/base (A0 C1 G2 T3, sixteen bases per
// uint32, first base in the low bits) + 1 bit/base that is set where the normalised base is 'N' (lower case folded to upper case,
// everything but A C G T -> N; the code of an N is 0), every read starting on a 32-base boundary.  k_pack2bit (mm_sketch.hip) produces
// it from ASCII on the GPU; this header produces the same words on the CPU, so that a caller whose bytes are in host memory anyway --
// the FASTA parser touches every base once while it drops the line breaks -- can ship 0.375 bytes per base over PCIe instead of 1
// (mm_reads_upload_packed).  Bit-identical to k_pack2bit for every byte value 0..255 (tests/test_pack2bit.py, tests/test_gpu_sketch.py).
//
// AVX-512BW + BMI2 (64 bases per step) or AVX2 + BMI2 (32 bases per step: compare against the four letters, movemasks, pdeps) where the
// CPU has them, a plain loop otherwise; chosen at run time.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace mmhost {

// one group of up to 32 bases -> (code word pair as one uint64, N mask); bases beyond n are absent (code 0, mask 0)
static inline void pack32_scalar(const unsigned char* s, size_t n, uint64_t& codes, uint32_t& nm) {
  uint64_t c = 0; uint32_t m = 0;
  for (size_t i = 0; i < n; i++) {
    const uint32_t ch = s[i] & 0xDFu;                                  // a-z -> A-Z (only letters can land on A/C/G/T)
    const bool ok = (ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T');
    c |= (uint64_t)(ok ? (((ch >> 1) ^ (ch >> 2)) & 3u) : 0u) << (2 * i);   // A0 C1 G2 T3
    m |= (ok ? 0u : 1u) << i;
  }
  codes = c; nm = m;
}

// n bases (any n) -> ceil(n/32) mask words and 2*ceil(n/32) code words; returns the number of N bases
static inline size_t pack2bit_scalar(const char* ascii, size_t n, uint32_t* bases2, uint32_t* nmask) {
  const unsigned char* s = (const unsigned char*)ascii;
  size_t nN = 0;
  for (size_t g = 0; g * 32 < n; g++) {
    uint64_t c; uint32_t m;
    pack32_scalar(s + g * 32, n - g * 32 < 32 ? n - g * 32 : 32, c, m);
    bases2[2 * g] = (uint32_t)c; bases2[2 * g + 1] = (uint32_t)(c >> 32); nmask[g] = m;
    nN += (size_t)__builtin_popcount(m);
  }
  return nN;
}

#if defined(__x86_64__)
__attribute__((target("avx2,bmi2")))
static inline size_t pack2bit_avx2(const char* ascii, size_t n, uint32_t* bases2, uint32_t* nmask) {
  const unsigned char* s = (const unsigned char*)ascii;
  const __m256i up = _mm256_set1_epi8((char)0xDF), cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
  size_t nN = 0;
  const size_t full = n / 32;
  for (size_t g = 0; g < full; g++) {
    const __m256i b = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(s + g * 32)), up);
    const __m256i isC = _mm256_cmpeq_epi8(b, cC), isG = _mm256_cmpeq_epi8(b, cG), isT = _mm256_cmpeq_epi8(b, cT);
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(b, cA), isC), _mm256_or_si256(isG, isT));
    // code bit 0 is set for C and T, bit 1 for G and T
    const uint32_t b0 = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(isC, isT));
    const uint32_t b1 = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(isG, isT));
    const uint32_t m = ~(uint32_t)_mm256_movemask_epi8(ok);
    const uint64_t c = _pdep_u64(b0, 0x5555555555555555ull) | _pdep_u64(b1, 0xAAAAAAAAAAAAAAAAull);
    bases2[2 * g] = (uint32_t)c; bases2[2 * g + 1] = (uint32_t)(c >> 32); nmask[g] = m;
    nN += (size_t)__builtin_popcount(m);
  }
  if (n % 32) {
    uint64_t c; uint32_t m;
    pack32_scalar(s + full * 32, n % 32, c, m);
    bases2[2 * full] = (uint32_t)c; bases2[2 * full + 1] = (uint32_t)(c >> 32); nmask[full] = m;
    nN += (size_t)__builtin_popcount(m);
  }
  return nN;
}
// AVX-512BW: the four compares of 64 bases come back as 64-bit masks (no movemask), four pdeps spread them into the code words
__attribute__((target("avx512f,avx512bw,bmi2")))
static inline size_t pack2bit_avx512(const char* ascii, size_t n, uint32_t* bases2, uint32_t* nmask) {
  const unsigned char* s = (const unsigned char*)ascii;
  const __m512i up = _mm512_set1_epi8((char)0xDF), cA = _mm512_set1_epi8('A'), cC = _mm512_set1_epi8('C'), cG = _mm512_set1_epi8('G'), cT = _mm512_set1_epi8('T');
  size_t nN = 0;
  const size_t full = n / 64;
  for (size_t g = 0; g < full; g++) {
    const __m512i b = _mm512_and_si512(_mm512_loadu_si512((const void*)(s + g * 64)), up);
    const uint64_t isA = _mm512_cmpeq_epi8_mask(b, cA), isC = _mm512_cmpeq_epi8_mask(b, cC), isG = _mm512_cmpeq_epi8_mask(b, cG), isT = _mm512_cmpeq_epi8_mask(b, cT);
    const uint64_t b0 = isC | isT, b1 = isG | isT, m = ~(isA | b0 | isG);
    const uint64_t c0 = _pdep_u64(b0 & 0xFFFFFFFFull, 0x5555555555555555ull) | _pdep_u64(b1 & 0xFFFFFFFFull, 0xAAAAAAAAAAAAAAAAull);
    const uint64_t c1 = _pdep_u64(b0 >> 32, 0x5555555555555555ull) | _pdep_u64(b1 >> 32, 0xAAAAAAAAAAAAAAAAull);
    memcpy(bases2 + 4 * g, &c0, 8); memcpy(bases2 + 4 * g + 2, &c1, 8); memcpy(nmask + 2 * g, &m, 8);
    nN += (size_t)__builtin_popcountll(m);
  }
  if (n % 64) nN += pack2bit_avx2(ascii + full * 64, n % 64, bases2 + 4 * full, nmask + 2 * full);
  return nN;
}
// 0 portable loop, 1 AVX2 + BMI2, 2 AVX-512BW + BMI2: the widest the CPU has; MASHMAP_HIP_PACK_ISA=scalar|avx2|avx512 narrows it (tests)
static inline int pack2bit_isa() {
  static const int isa = [] {
    int best = 0;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2")) best = 1;
    if (best == 1 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")) best = 2;
    if (const char* e = getenv("MASHMAP_HIP_PACK_ISA")) {
      const int want = !strcmp(e, "scalar") ? 0 : !strcmp(e, "avx2") ? 1 : !strcmp(e, "avx512") ? 2 : best;
      if (want < best) best = want;
    }
    return best;
  }();
  return isa;
}
#endif

static inline size_t pack2bit(const char* ascii, size_t n, uint32_t* bases2, uint32_t* nmask) {
#if defined(__x86_64__)
  const int isa = pack2bit_isa();
  if (isa == 2) return pack2bit_avx512(ascii, n, bases2, nmask);
  if (isa == 1) return pack2bit_avx2(ascii, n, bases2, nmask);
#endif
  return pack2bit_scalar(ascii, n, bases2, nmask);
}

// ---- line breaks.  A FASTA body is sequence bytes with a '\n' every so many columns; the parser wants them gone before the packer sees
// the bytes.  Per-line memchr + memcpy costs more than the packing at 60..100 columns, so: strip_newlines() compacts a range with
// AVX-512 VBMI2 (vpcompressb, 64 bytes per step) where the CPU has it, count_newlines() counts with AVX2 (the sizing pass of the
// split-record parser, seq_parse.hpp); plain loops otherwise.
static inline size_t strip_newlines_scalar(const char* p, size_t n, char* dst) {
  const char* e = p + n; char* d = dst;
  while (p < e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    const size_t m = (size_t)((nl ? nl : e) - p);
    memcpy(d, p, m); d += m;
    p = nl ? nl + 1 : e;
  }
  return (size_t)(d - dst);
}
static inline size_t count_newlines_scalar(const char* p, size_t n) {
  size_t c = 0;
  for (size_t i = 0; i < n; i++) c += p[i] == '\n';
  return c;
}
#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw,avx512vbmi2")))
static inline size_t strip_newlines_vbmi2(const char* p, size_t n, char* dst) {     // dst needs room for n + 64 bytes
  const __m512i nlv = _mm512_set1_epi8('\n');
  char* d = dst;
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    const __m512i v = _mm512_loadu_si512((const void*)(p + i));
    const __mmask64 keep = _mm512_cmpneq_epi8_mask(v, nlv);
    _mm512_storeu_si512((void*)d, _mm512_maskz_compress_epi8(keep, v));
    d += __builtin_popcountll((unsigned long long)keep);
  }
  for (; i < n; i++) if (p[i] != '\n') *d++ = p[i];
  return (size_t)(d - dst);
}
__attribute__((target("avx2")))
static inline size_t count_newlines_avx2(const char* p, size_t n) {
  const __m256i nlv = _mm256_set1_epi8('\n');
  size_t c = 0, i = 0;
  for (; i + 32 <= n; i += 32) c += (size_t)__builtin_popcount((unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(p + i)), nlv)));
  for (; i < n; i++) c += p[i] == '\n';
  return c;
}
static inline bool have_vbmi2() {
  static const bool v = __builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("avx512bw") && pack2bit_isa() == 2;   // MASHMAP_HIP_PACK_ISA narrows this too
  return v;
}
#endif
// scan_to_header: from q (a line start that is not a header line) to the next header line -- a '>' right behind a '\n' -- or to e; returns
// where it stopped and adds the line breaks of [q, there) to *newlines.  One sweep: the sizing pass of the split-record parser reads every
// byte of a chromosome once.
static inline const char* scan_to_header_scalar(const char* q, const char* e, size_t* newlines) {
  const char* h = q;
  for (;;) {
    h = (const char*)memchr(h, '>', (size_t)(e - h));
    if (!h) { h = e; break; }
    if (h > q && h[-1] == '\n') break;
    h++;
  }
  *newlines += count_newlines_scalar(q, (size_t)(h - q));
  return h;
}
#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw")))
static inline const char* scan_to_header_avx512(const char* q, const char* e, size_t* newlines) {
  const __m512i nlv = _mm512_set1_epi8('\n'), gtv = _mm512_set1_epi8('>');
  size_t c = 0;
  const char* p = q;
  for (; p + 64 <= e; p += 64) {
    const __m512i v = _mm512_loadu_si512((const void*)p);
    const uint64_t mnl = _mm512_cmpeq_epi8_mask(v, nlv);
    uint64_t mgt = _mm512_cmpeq_epi8_mask(v, gtv);
    if (mgt) {
      // a '>' is a header's first byte iff the byte before it is a line break (inside the block: the bit below; at bit 0: the byte before the block)
      uint64_t cand = mgt & (mnl << 1);
      if ((mgt & 1ull) && p > q && p[-1] == '\n') cand |= 1ull;
      if (cand) {
        const int i = __builtin_ctzll(cand);
        c += (size_t)__builtin_popcountll(mnl & ((1ull << i) - 1ull));
        *newlines += c;
        return p + i;
      }
    }
    c += (size_t)__builtin_popcountll(mnl);
  }
  *newlines += c;
  for (; p < e; p++) {                                     // the last partial block
    if (*p == '\n') (*newlines)++;
    else if (*p == '>' && p > q && p[-1] == '\n') return p;
  }
  return e;
}
#endif
static inline size_t strip_newlines(const char* p, size_t n, char* dst) {
#if defined(__x86_64__)
  if (have_vbmi2()) return strip_newlines_vbmi2(p, n, dst);
#endif
  return strip_newlines_scalar(p, n, dst);
}
static inline const char* scan_to_header(const char* q, const char* e, size_t* newlines) {
#if defined(__x86_64__)
  if (pack2bit_isa() == 2) return scan_to_header_avx512(q, e, newlines);
#endif
  return scan_to_header_scalar(q, e, newlines);
}
static inline size_t count_newlines(const char* p, size_t n) {
#if defined(__x86_64__)
  if (pack2bit_isa() >= 1) return count_newlines_avx2(p, n);
#endif
  return count_newlines_scalar(p, n);
}

// Streaming form for input that arrives in pieces (FASTA lines): feed() any number of byte ranges, finish() once.  Groups of 32 bases
// are packed as soon as they are complete; at most 31 bases wait in `carry`.
struct Pack2bitStream {
  uint32_t* b2; uint32_t* nm; size_t groups = 0, nN = 0; unsigned char carry[32]; size_t nCarry = 0;
  Pack2bitStream(uint32_t* bases2, uint32_t* nmask) : b2(bases2), nm(nmask) {}
  void feed(const char* p, size_t n) {
    if (nCarry) {
      const size_t take = 32 - nCarry < n ? 32 - nCarry : n;
      memcpy(carry + nCarry, p, take); nCarry += take; p += take; n -= take;
      if (nCarry < 32) return;
      nN += pack2bit((const char*)carry, 32, b2 + 2 * groups, nm + groups); groups++; nCarry = 0;
    }
    const size_t full = n / 32;
    if (full) { nN += pack2bit(p, full * 32, b2 + 2 * groups, nm + groups); groups += full; }
    nCarry = n - full * 32;
    if (nCarry) memcpy(carry, p + full * 32, nCarry);
  }
  size_t finish() {                                                    // returns the number of N bases of the whole read
    if (nCarry) { nN += pack2bit((const char*)carry, nCarry, b2 + 2 * groups, nm + groups); groups++; nCarry = 0; }
    return nN;
  }
};

}  // namespace mmhost

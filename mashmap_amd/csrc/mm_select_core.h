// mashmap_amd/csrc/mm_select_core.h -- doL2Mapping's best-first walk over ONE fragment's L1 candidates, in integers.
//
//   Map::mapSingleQueryFrag   reference groups + std::make_heap       src/map/include/computeMap.hpp:774-796
//   Map::doL2Mapping          best-first, ANI cut-off, acceptance     src/map/include/computeMap.hpp:1182-1267
//
// Shared by the kernel (k_l2_select, mm_select.hip) and by the CPU test harness (tests/hostlogic/hostlogic.cpp), which runs it
// beside the float-based host replay on the reference's own L1/L2 integers.  The two float decisions of the walk arrive as rows
// of the host tables (mmhost::replayTables): acc[shared] (:1221-1222) and cut[best] (:1192-1202) for this fragment's Q.sketchSize.
#pragma once
#include "mm_heap.h"
#include "../../include/mashmap_hip.h"

// cl: the fragment's nC candidates in emission order; heap: nC ints of scratch; l2First/l2Num: per candidate (same indexing as cl),
// first locus in l2 and count.  emit(locus) is called for every reported locus, in the reference's push order; returns their number.
template <class Emit>
MM_HD int mm_select_fragment(int nC, const mm_l1_candidate* cl, int32_t* heap, const int64_t* l2First, const int32_t* l2Num,
                             const mm_l2_locus* l2, const int32_t* refGroup, int skipPrefix, int hg, int Qs,
                             const uint8_t* acc, const int16_t* cut, Emit emit) {
  int n = 0;
  auto less = [cl](int32_t x, int32_t y) { return cl[x].intersectionSize < cl[y].intersectionSize; };   // L1_locus_intersection_cmp (:71)
  int gb = 0;
  while (gb < nC) {                                               // one doL2Mapping call per reference group (skip_prefix), :776-796
    int ge = nC;
    if (skipPrefix) { const int g = refGroup[cl[gb].seqId]; ge = gb; while (ge < nC && refGroup[cl[ge].seqId] == g) ge++; }
    const int len0 = ge - gb;
    int32_t* h = heap + gb;
    for (int i = 0; i < len0; i++) h[i] = gb + i;
    if (hg) mm_make_heap(h, len0, less);
    int best = 0, endIdx = len0, it = 0;
    while (it != endIdx) {
      const int c = h[it];
      if (hg && cl[c].intersectionSize < (int)cut[best]) break;   // ANI cut-off against the best locus reported so far (:1192-1202)
      const int64_t l0 = l2First[c]; const int ln = l2Num[c];
      for (int i = 0; i < ln; i++) {
        const mm_l2_locus& L = l2[l0 + i];
        const int shared = L.sharedSketchSize;
        if (shared <= Qs && acc[shared]) {                         // identity (or its upper bound) reaches percentageIdentity (:1221-1222)
          best = shared > best ? shared : best;
          emit(L);
          n++;
        }
      }
      if (hg) { mm_pop_heap(h, endIdx, less); endIdx--; } else it++;
    }
    gb = ge;
  }
  return n;
}

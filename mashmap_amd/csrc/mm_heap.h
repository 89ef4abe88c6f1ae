// mashmap_amd/csrc/mm_heap.h -- binary max-heap over an index array with the exact element movements of libstdc++'s
// std::make_heap / std::pop_heap (bits/stl_heap.h: __push_heap, __adjust_heap, __make_heap, __pop_heap).
//
// Map::mapSingleQueryFrag heaps a fragment's L1 candidates by intersectionSize (std::make_heap, computeMap.hpp:791) and
// doL2Mapping consumes them best-first with std::pop_heap (:1256).  Candidates with EQUAL intersectionSize come out in whatever
// order that heap implementation yields, and the order decides which of them still pass the moving ANI cut-off (:1192-1202).
// The reference is built against libstdc++, so its order is reproduced here step for step; tests/hostlogic/heap_check.cpp
// compares this header with std::make_heap / std::pop_heap exhaustively on small arrays with ties (CPU test suite).
// Compiles for the host (g++) and for the device (hipcc).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define MM_HD __host__ __device__ __forceinline__
#else
#define MM_HD inline
#endif

// less(a, b): "a orders before b" == comp(a, b) of the std:: calls (a max-heap keeps the element that is not less than any at the front)
template <class Less>
MM_HD void mm_heap_push(int32_t* first, int hole, int top, int32_t value, Less less) {
  int parent = (hole - 1) / 2;
  while (hole > top && less(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

template <class Less>
MM_HD void mm_heap_adjust(int32_t* first, int hole, int len, int32_t value, Less less) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(first[child], first[child - 1])) child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  mm_heap_push(first, hole, top, value, less);
}

template <class Less>
MM_HD void mm_make_heap(int32_t* first, int len, Less less) {
  if (len < 2) return;
  int parent = (len - 2) / 2;
  while (true) {
    const int32_t v = first[parent];
    mm_heap_adjust(first, parent, len, v, less);
    if (parent == 0) return;
    parent--;
  }
}

// std::pop_heap(first, first + len): the front moves to first[len - 1], the rest is a heap again
template <class Less>
MM_HD void mm_pop_heap(int32_t* first, int len, Less less) {
  if (len < 2) return;
  const int32_t v = first[len - 1];
  first[len - 1] = first[0];
  mm_heap_adjust(first, 0, len - 1, v, less);
}

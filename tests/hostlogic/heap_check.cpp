// tests/hostlogic/heap_check.cpp -- mm_heap.h against std::make_heap / std::pop_heap (libstdc++), which is what the reference's
// doL2Mapping order rests on (computeMap.hpp:791,1256).  Exhaustive over all key assignments with ties for n <= 7, random beyond.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../mashmap_amd/csrc/mm_heap.h"

struct Cand { int id, isz; };
static long checks = 0;
static bool one(const std::vector<int>& keys) {
  const int n = (int)keys.size();
  std::vector<Cand> a(n);
  for (int i = 0; i < n; i++) a[i] = Cand{i, keys[i]};
  std::vector<int32_t> idx(n);
  for (int i = 0; i < n; i++) idx[i] = i;
  auto cmpS = [](const Cand& x, const Cand& y) { return x.isz < y.isz; };
  auto less = [&keys](int32_t x, int32_t y) { return keys[x] < keys[y]; };
  std::make_heap(a.begin(), a.end(), cmpS);
  mm_make_heap(idx.data(), n, less);
  for (int len = n; len >= 1; len--) {
    for (int i = 0; i < len; i++) if (a[i].id != idx[i]) return false;
    std::pop_heap(a.begin(), a.begin() + len, cmpS);
    mm_pop_heap(idx.data(), len, less);
    checks++;
  }
  return true;
}
int main() {
  for (int n = 1; n <= 7; n++) {
    std::vector<int> keys(n, 0);
    const int vals = n < 4 ? n : 4;                 // keys in [0, vals): plenty of ties
    long total = 1; for (int i = 0; i < n; i++) total *= vals;
    for (long code = 0; code < total; code++) {
      long c = code; for (int i = 0; i < n; i++) { keys[i] = (int)(c % vals); c /= vals; }
      if (!one(keys)) { printf("MISMATCH n=%d code=%ld\n", n, code); return 1; }
    }
  }
  srand(12345);
  for (int it = 0; it < 20000; it++) {
    const int n = 1 + rand() % 200;
    std::vector<int> keys(n);
    const int range = 1 + rand() % 12;
    for (auto& k : keys) k = rand() % range;
    if (!one(keys)) { printf("MISMATCH random it=%d\n", it); return 1; }
  }
  printf("heap order identical to libstdc++ on %ld pops\n", checks);
  return 0;
}

// tests/hostlogic/parse_check.cpp -- TEST HARNESS for mashmap_amd/host/seq_parse.hpp (no GPU): parses the files given on the command
// line with a window size and thread count and prints, per record, "name<TAB>length<TAB>fnv1a64 of the sequence".
#include <cstdio>
#include <cstdlib>
#include "../../mashmap_amd/host/seq_parse.hpp"

// "pool": mmhost::WorkerPool under stress -- thousands of short runs of varying width, every index of every run exactly once, runs
// never leaking into each other; prints "pool ok <runs> <tasks>" or the first violation
static int pool_check(unsigned threads, int runs) {
  mmhost::WorkerPool pool(threads);
  std::vector<std::atomic<int>> hits(4096);
  unsigned long long total = 0;
  for (int r = 0; r < runs; r++) {
    const unsigned n = (unsigned)(1 + (r * 2654435761u >> 7) % (r % 5 == 0 ? 4096 : 3 * threads));
    for (unsigned i = 0; i < n; i++) hits[i].store(0);
    std::atomic<unsigned long long> sum(0);
    pool.run(n, [&](unsigned t) { hits[t].fetch_add(1); sum.fetch_add((unsigned long long)t + 1); if ((t & 63) == 0) std::this_thread::yield(); });
    for (unsigned i = 0; i < n; i++) if (hits[i].load() != 1) { printf("pool FAIL run %d index %u executed %d times\n", r, i, hits[i].load()); return 1; }
    if (sum.load() != (unsigned long long)n * (n + 1) / 2) { printf("pool FAIL run %d sum\n", r); return 1; }
    total += n;
  }
  printf("pool ok %d %llu\n", runs, total);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 4 && std::string(argv[1]) == "pool") return pool_check((unsigned)atoi(argv[2]), atoi(argv[3]));
  if (argc == 2 && std::string(argv[1]) == "cpus") { printf("%u\n", mmhost::availableCpus()); return 0; }   // what the stage widths are capped by
  if (argc < 4) return 2;
  const size_t window = (size_t)atol(argv[1]); const unsigned threads = (unsigned)atoi(argv[2]);
  std::string prefix; int a = 3; bool packed = false;
  if (std::string(argv[a]) == "--packed") { packed = true; a++; }      // the reader packs (2-bit codes + N mask); the hash is of the normalised sequence they encode
  if (std::string(argv[a]) == "--prefix") { prefix = argv[a + 1]; a += 2; }
  std::vector<std::string> files;
  for (int i = a; i < argc; i++) files.push_back(argv[i]);
  mmhost::BatchReader rd(files, window, threads, {}, prefix, nullptr, nullptr, packed);
  mmhost::ParsedBatch b;
  size_t batches = 0;
  while (rd.next(b)) {
    batches++;
    for (size_t r = 0; r < b.size(); r++) {
      unsigned long long h = 1469598103934665603ull;
      if (!packed) for (int64_t i = b.offs[r]; i < b.offs[r + 1]; i++) { h ^= (unsigned char)b.bases[i]; h *= 1099511628211ull; }
      else {
        const int64_t len = b.offs[r + 1] - b.offs[r], p0 = b.packOffs[r];
        const int64_t span = (len + 31) / 32 * 32;                      // a record's own words; the next record may start later (gaps between the threads' pieces)
        if (b.lens[r] != len || p0 % 32 || b.packOffs[r + 1] - p0 < span || b.packEnd(r, r + 1) != p0 + span) { printf("bad packed layout at record %zu\n", r); return 1; }
        bool anyN = false;
        for (int64_t i = 0; i < span; i++) {
          const int64_t g = p0 + i;
          const unsigned code = (b.bases2()[g >> 4] >> (2 * (g & 15))) & 3u, isN = (b.nmask()[g >> 5] >> (g & 31)) & 1u;
          if (i >= len) { if (code || isN) { printf("padding not zero at record %zu\n", r); return 1; } continue; }
          if (isN && code) { printf("N with a code at record %zu\n", r); return 1; }
          anyN |= isN != 0;
          h ^= (unsigned char)(isN ? 'N' : "ACGT"[code]); h *= 1099511628211ull;
        }
        if ((b.hasN[r] != 0) != anyN) { printf("hasN wrong at record %zu\n", r); return 1; }
      }
      printf("%s\t%lld\t%llu\n", b.names[r].c_str(), (long long)(b.offs[r + 1] - b.offs[r]), h);
    }
  }
  rd.release(b);
  fprintf(stderr, "batches %zu split %zu\n", batches, rd.splitWindows());
  return 0;
}

// tests/hostlogic/parse_check.cpp -- TEST HARNESS for mashmap_amd/host/seq_parse.hpp (no GPU): parses the files given on the command
// line with a window size and thread count and prints, per record, "name<TAB>length<TAB>fnv1a64 of the sequence".
#include <cstdio>
#include <cstdlib>
#include "../../mashmap_amd/host/seq_parse.hpp"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const size_t window = (size_t)atol(argv[1]); const unsigned threads = (unsigned)atoi(argv[2]);
  std::string prefix; int a = 3;
  if (std::string(argv[3]) == "--prefix") { prefix = argv[4]; a = 5; }
  std::vector<std::string> files;
  for (int i = a; i < argc; i++) files.push_back(argv[i]);
  mmhost::BatchReader rd(files, window, threads, {}, prefix);
  mmhost::ParsedBatch b;
  size_t batches = 0;
  while (rd.next(b)) {
    batches++;
    for (size_t r = 0; r < b.size(); r++) {
      unsigned long long h = 1469598103934665603ull;
      for (int64_t i = b.offs[r]; i < b.offs[r + 1]; i++) { h ^= (unsigned char)b.bases[i]; h *= 1099511628211ull; }
      printf("%s\t%lld\t%llu\n", b.names[r].c_str(), (long long)(b.offs[r + 1] - b.offs[r]), h);
    }
  }
  rd.release(b);
  fprintf(stderr, "batches %zu\n", batches);
  return 0;
}

// tests/hostlogic/commonfunc_check.cpp -- TEST HARNESS for the commonFunc.hpp-level seams (mashmap_amd/host/skch_commonfunc.hpp):
// a caller written against skch::CommonFunc::sketchSequence / addMinmers with the reference's signatures (commonFunc.hpp:183,302).
// Standalone build: the repository's own declarations.  With -DMASHMAP_HIP_REFERENCE_TREE (and reference_tree/ first on the include
// path) the same source compiles against the reference's commonFunc.hpp overlaid by reference_tree/map/include/commonFunc.hpp.
// usage: commonfunc_check K W S file      (one sequence per line); prints "S|M line hash wpos wpos_end seqId strand"
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>
#ifdef MASHMAP_HIP_REFERENCE_TREE
#include "map/include/base_types.hpp"
#include "map/include/commonFunc.hpp"
#else
#include "../../mashmap_amd/host/skch_commonfunc.hpp"
#endif

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int K = atoi(argv[1]), W = atoi(argv[2]), S = atoi(argv[3]);
  std::ifstream in(argv[4]);
  std::string line; int no = 0;
  while (std::getline(in, line)) {
    std::vector<skch::MinmerInfo> sk(3, skch::MinmerInfo{1, 2, 3, 4, 5});     // sketchSequence replaces the content
    std::string a = line;
    skch::CommonFunc::sketchSequence(sk, &a[0], (skch::offset_t)a.size(), K, 4, S, (skch::seqno_t)(100 + no));
    for (const auto& m : sk) printf("S %d %llu %d %d %d %d\n", no, (unsigned long long)m.hash, m.wpos, m.wpos_end, m.seqId, (int)m.strand);
    std::vector<skch::MinmerInfo> mi;
    std::string b = line;
    skch::CommonFunc::addMinmers(mi, &b[0], (skch::offset_t)b.size(), K, W, 4, S, (skch::seqno_t)(7 + no));
    for (const auto& m : mi) printf("M %d %llu %d %d %d %d\n", no, (unsigned long long)m.hash, m.wpos, m.wpos_end, m.seqId, (int)m.strand);
    printf("N %d %s\n", no, a == b ? a.substr(0, 40).c_str() : "normalisation differs");                 // seq is normalised in place
    no++;
  }
  return 0;
}

// Drop-in overlay for src/map/include/commonFunc.hpp inside the reference tree: everything the reference's header
// defines stays (getHash, reverseComplement, makeUpperCaseAndValidDNA, split, getReferenceSize ...), except that
// skch::CommonFunc::sketchSequence (:183) and skch::CommonFunc::addMinmers (:302) are the versions backed by libmashmap_hip.so
// (skch_commonfunc.hpp); the reference's own CPU templates remain reachable as sketchSequence_reference_cpu / addMinmers_reference_cpu.
// Needs this directory before the reference's src/ on the include path (#include_next continues the search behind it).
#pragma once
#ifndef MASHMAP_HIP_REFERENCE_TREE
#define MASHMAP_HIP_REFERENCE_TREE 1
#endif
#define sketchSequence sketchSequence_reference_cpu
#define addMinmers addMinmers_reference_cpu
#include_next "map/include/commonFunc.hpp"
#undef sketchSequence
#undef addMinmers
#include "skch_commonfunc.hpp"
namespace skch {
namespace CommonFunc {
template <typename T>
inline void sketchSequence(std::vector<T>& minmerIndex, char* seq, offset_t len, int kmerSize, int alphabetSize, int sketchSize, seqno_t seqCounter) {
  hipseam::sketchSequence(minmerIndex, seq, len, kmerSize, alphabetSize, sketchSize, seqCounter);
}
template <typename T>
inline void addMinmers(std::vector<T>& minmerIndex, char* seq, offset_t len, int kmerSize, int windowSize, int alphabetSize, int sketchSize, seqno_t seqCounter) {
  hipseam::addMinmers(minmerIndex, seq, len, kmerSize, windowSize, alphabetSize, sketchSize, seqCounter);
}
}  // namespace CommonFunc
}  // namespace skch

// mashmap_amd/csrc/mm_exchange_plan.h -- layout of the one exchange step of a multi-GPU run (host code, no device types).
//
// The all-gatherv of the candidate mappings (mm_comm.hip) is: (1) an all-gather of one record count per rank, (2) slot r of the gathered
// buffer = rank r's records, slots in rank order without padding -- with reads sharded in contiguous blocks, rank-major is input
// order --, (3) `world` broadcasts, root r sending its count[r] records into everybody's slot r (empty slots are
// skipped by every rank alike, so the collectives stay matched).  This header is step (2): shared by the RCCL path of
// libmashmap_hip.so and, through libmashmap_host.so (mmh_exchange_plan), by the CPU tests that run the same protocol over gloo.
#pragma once
#include <stddef.h>
#include <stdint.h>

// disp[r] = first record of rank r's slot, disp[world] = total number of gathered records
static inline uint64_t mm_exchange_place(const uint64_t* counts, int world, uint64_t* disp) {
  uint64_t at = 0;
  for (int r = 0; r < world; r++) { disp[r] = at; at += counts[r]; }
  disp[world] = at;
  return at;
}

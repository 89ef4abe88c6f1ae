// mashmap_amd/csrc/mm_winnow.h -- device winnowing (mm_winnow.hip) as seen by mm_index.hip
#pragma once
#include "mm_internal.h"

#define WN_CARRY ((int32_t)0x80000000)     // "run was already open at the tile's first window": start comes from the previous tile

struct WnOpenRun { uint64_t hash; int32_t start; int32_t sum; };   // a sketch member still open at the end of a tile

struct WinnowBuffers {       // grow-only device scratch shared by the contigs of one mm_index_build call
  DevBuf blockCnt, blockOff, cPos, cHash, cSt;
  DevBuf out, outCount, outOff, open, openCount, status, dense;
  DevBuf redoList, out2, outCount2, open2, openCount2, status2;
  // page-locked landing area for a contig's records and open lists (hundreds of megabytes per contig at human scale): the copy down
  // runs at the link's rate instead of through the runtime's pageable staging, and the host vectors are filled from it in one pass
  // (a vector sized first is zero-filled first)
  void* hStage = nullptr; size_t hStageBytes = 0;
  void* host(size_t bytes) {                     // nullptr when page-locking fails: the caller copies into pageable memory as before
    if (bytes <= hStageBytes) return hStage;
    if (hStage) { (void)hipHostFree(hStage); hStage = nullptr; hStageBytes = 0; }
    const size_t cap = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&hStage, cap, hipHostMallocDefault) != hipSuccess) { hStage = nullptr; (void)hipGetLastError(); return nullptr; }
    hStageBytes = cap;
    return hStage;
  }
  void release() {
    if (hStage) { (void)hipHostFree(hStage); hStage = nullptr; hStageBytes = 0; }
    DevBuf* all[] = {&blockCnt, &blockOff, &cPos, &cHash, &cSt, &out, &outCount, &outOff, &open, &openCount, &status, &dense,
                     &redoList, &out2, &outCount2, &open2, &openCount2, &status2};
    for (DevBuf* b : all) b->release();
  }
};

// records in the reference's emission order (wpos == WN_CARRY where the run started before its tile), per-tile record counts,
// and every tile's open list (s slots per tile) for the stitching pass
int mm_winnow_contig_device(mm_ctx* c, WinnowBuffers& B, const uint64_t* dH, const int8_t* dS, int64_t nPos, int len,
                            std::vector<mm_minmer>& records, std::vector<int32_t>& tileCount, std::vector<WnOpenRun>& openRuns,
                            std::vector<int32_t>& openCount);

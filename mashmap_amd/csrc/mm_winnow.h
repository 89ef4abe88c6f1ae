// mashmap_amd/csrc/mm_winnow.h -- device winnowing (mm_winnow.hip) as seen by mm_index.hip
#pragma once
#include "mm_internal.h"

#define WN_CARRY ((int32_t)0x80000000)     // "run was already open at the tile's first window": start comes from the previous tile

struct WnOpenRun { uint64_t hash; int32_t start; int32_t sum; };   // a sketch member still open at the end of a tile

struct WinnowBuffers {       // grow-only device scratch shared by the contigs of one mm_index_build call
  DevBuf blockCnt, blockOff, cPos, cHash, cSt;
  DevBuf out, outCount, outOff, open, openCount, status, dense;
  DevBuf redoList, out2, outCount2, open2, openCount2, status2;
  void release() {
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

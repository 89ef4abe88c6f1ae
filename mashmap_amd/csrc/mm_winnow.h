// mashmap_amd/csrc/mm_winnow.h -- device winnowing (mm_winnow.hip) as seen by mm_index.hip
#pragma once
#include "mm_internal.h"

#define WN_CARRY ((int32_t)0x80000000)     // "run was already open at the tile's first window": start comes from the previous tile

struct WnOpenRun { uint64_t hash; int32_t start; int32_t sum; };   // a sketch member still open at the end of a tile

struct WinnowBuffers {       // grow-only device scratch shared by the contigs of one mm_index_build call
  DevBuf blockCnt, blockOff, cPos, cHash, cSt;
  DevBuf out, outCount, outOff, open, openCount, status, dense;
  DevBuf redoList, out2, outCount2, open2, openCount2, status2;
  DevBuf skScratch;          // k_winnow_tiles<.., GSK>: the window's sketch per tile slot, for sketches LDS does not hold
  // page-locked landing area for a contig's records and open lists (hundreds of megabytes per contig at human scale): the copy down
  // runs at the link's rate instead of through the runtime's pageable staging, and the host vectors are filled from it in one pass
  // (a vector sized first is zero-filled first)
  // Two of them, taken in turns: the host thread that finishes contig i copies its records out of one while the device fills the other
  // with contig i+1's (the caller waits for the copy-out of contig i-2 before it hands a buffer to the device again).
  void* hStage[2] = {nullptr, nullptr}; size_t hStageBytes[2] = {0, 0}; int turn = 0;
  void* host(size_t bytes) {                     // nullptr when page-locking fails: the caller copies into pageable memory as before
    void*& h = hStage[turn]; size_t& hb = hStageBytes[turn];
    if (bytes <= hb) return h;
    if (h) { (void)hipHostFree(h); h = nullptr; hb = 0; }
    const size_t cap = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&h, cap, hipHostMallocDefault) != hipSuccess) { h = nullptr; (void)hipGetLastError(); return nullptr; }
    hb = cap;
    return h;
  }
  void release() {
    for (int i = 0; i < 2; i++) if (hStage[i]) { (void)hipHostFree(hStage[i]); hStage[i] = nullptr; hStageBytes[i] = 0; }
    DevBuf* all[] = {&blockCnt, &blockOff, &cPos, &cHash, &cSt, &out, &outCount, &outOff, &open, &openCount, &status, &dense,
                     &redoList, &out2, &outCount2, &open2, &openCount2, &status2, &skScratch};
    for (DevBuf* b : all) b->release();
  }
};

// records in the reference's emission order (wpos == WN_CARRY where the run started before its tile), per-tile record counts,
// and every tile's open list (s slots per tile) for the stitching pass
// staged (may be null): when the contig's records and open lists have landed in the page-locked buffer of this turn and no tile had to
// be redone, they are LEFT there -- `records` / `openRuns` stay empty and the caller copies them out (WnStaged::take) off the device's
// critical path
struct WnStaged {
  const unsigned char* hs = nullptr; size_t total = 0, recPad = 0, nRuns = 0;
  void take(std::vector<mm_minmer>& records, std::vector<WnOpenRun>& openRuns) const {
    if (!hs) return;
    const mm_minmer* r0 = (const mm_minmer*)hs; const WnOpenRun* o0 = (const WnOpenRun*)(hs + recPad);
    records.assign(r0, r0 + total); openRuns.assign(o0, o0 + nRuns);
  }
};
int mm_winnow_contig_device(mm_ctx* c, WinnowBuffers& B, const uint64_t* dH, const int8_t* dS, int64_t nPos, int len,
                            std::vector<mm_minmer>& records, std::vector<int32_t>& tileCount, std::vector<WnOpenRun>& openRuns,
                            std::vector<int32_t>& openCount, WnStaged* staged = nullptr);

// mashmap_amd/csrc/mm_comm.hip -- the one exchange step of a multi-GPU run: all-gatherv of the candidate mappings over RCCL (xGMI).
//
// The reference maps every read in one address space; chaining and the per-read filters need only that read's fragments
// (computeMap.hpp:679-697), but the one-to-one filter (:358-405) and the single output stream need every read's mappings in one
// place.  Here reads are sharded over GPUs in contiguous blocks with the index replicated, fragments never
// interact before the CPU filters, so the data path has exactly one collective: after mm_map_fragments every rank contributes its
// batch's mm_mapping records (48 bytes per reported locus, ~1 per 5 kbp of reads) and receives everybody's, rank-major -- which, with
// contiguous read blocks, is input order.
//
//   one process per GPU :  mm_comm_unique_id (rank 0) -> ship the 128 bytes -> mm_comm_init_rank on every rank -> mm_allgatherv_mappings
//   one process, n GPUs :  mm_comm_init_local(ctxs, n) -> mm_allgatherv_mappings_local(ctxs, n)     (skch::Map, MASHMAP_HIP_DEVICES)
//
// The all-gatherv is one RCCL group of `world` broadcasts (root r sends its count[r] records into everybody's slot r): no padding,
// no staging copy.  Counts travel first (ncclAllGather of one uint64 per rank; the host needs them to size and place the slots).
// RCCL is opened with dlopen on first use: single-GPU users of libmashmap_hip.so never load it.  Contexts of a local group that
// share a device (two contexts on one GPU) cannot form an RCCL communicator; they exchange by device-to-device copies instead.
#include "mm_internal.h"
#include "mm_exchange_plan.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;        // optional
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;        // optional
  std::string err, path;                               // path: the file the entry points were bound from
};
Rccl g_rccl;
std::mutex g_rcclMu;

Rccl* rccl_open(std::string& err) {
  std::lock_guard<std::mutex> lk(g_rcclMu);
  if (g_rccl.h) return &g_rccl;
  if (!g_rccl.err.empty()) { err = g_rccl.err; return nullptr; }
  // A process that already has an RCCL mapped (torch ships its own librccl.so: bench.py's ranks run dist.init_process_group("nccl") beside
  // this library's communicator) must not get a second copy of the library with its own global state: take the mapped one first
  // (RTLD_NOLOAD), load one only if there is none.  MASHMAP_HIP_RCCL=path pins a file; MM_DEBUG / MASHMAP_HIP_TIMING log what was bound.
  const char* names[] = {getenv("MASHMAP_HIP_RCCL"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr; const char* how = "already mapped";
  for (const char* n : names) { if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break; }
  if (!h) { how = "loaded"; for (const char* n : names) { if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break; } }
  if (!h) { g_rccl.err = std::string("cannot load RCCL (librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); err = g_rccl.err; return nullptr; }
  {
    Dl_info di; void* f = dlsym(h, "ncclGetUniqueId");
    g_rccl.path = (f && dladdr(f, &di) && di.dli_fname) ? di.dli_fname : "?";
    if (getenv("MM_DEBUG") || getenv("MASHMAP_HIP_TIMING")) fprintf(stderr, "[mm] RCCL bound to %s (%s)\n", g_rccl.path.c_str(), how);
  }
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) ok = false; return p; };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(h, "ncclCommAbort");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
  if (!ok) { g_rccl.err = "RCCL library lacks a required entry point"; err = g_rccl.err; dlclose(h); return nullptr; }
  g_rccl.h = h;
  return &g_rccl;
}

#define MM_NCCL(ctx, R, call)                                                                    \
  do {                                                                                           \
    ncclResult_t r__ = (call);                                                                   \
    if (r__ != ncclSuccess) { (ctx)->err = std::string(#call) + ": " + (R)->GetErrorString(r__); return MM_ERR_DEVICE; } \
  } while (0)

// slots of the gathered buffer from the per-rank counts
void place(mm_ctx* c) {
  const int world = (int)c->gatherCounts.size();
  std::vector<uint64_t> cnt(c->gatherCounts.begin(), c->gatherCounts.end()), disp((size_t)world + 1, 0);
  c->nGathered = (size_t)mm_exchange_place(cnt.data(), world, disp.data());
  c->gatherDisp.assign(disp.begin(), disp.end());
}

// the `world` broadcasts of one rank (to be called between GroupStart / GroupEnd); `mine` = this rank's records
struct ErrSink { std::string err; };   // error text of the exchange thread (c->err belongs to the caller's thread)

template <class E>
int issue_broadcasts(mm_ctx* c, E* e, Rccl* R, const void* mine, hipStream_t stream) {
  const int world = c->commWorld;
  for (int r = 0; r < world; r++) {
    const size_t bytes = c->gatherCounts[r] * sizeof(mm_mapping);
    if (!bytes) continue;
    char* slot = (char*)c->dGathered.p + c->gatherDisp[r] * sizeof(mm_mapping);
    MM_NCCL(e, R, R->Broadcast(r == c->commRank ? mine : (const void*)slot, slot, bytes, ncclChar, r, (ncclComm_t)c->comm, stream));
  }
  return MM_OK;
}

// the exchange of one rank: counts (all-gather), slots, `world` grouped broadcasts; returns with `stream` drained
template <class E>
int exchange(mm_ctx* c, E* e, Rccl* R, const void* mine, size_t nMine, hipStream_t stream) {
  MM_HIP(e, hipSetDevice(c->device));
  const int world = c->commWorld;
  unsigned long long cnt = nMine;
  unsigned long long* dC = c->dCommCounts.as<unsigned long long>();
  MM_HIP(e, hipMemcpyAsync(dC + world, &cnt, 8, hipMemcpyHostToDevice, stream));
  MM_NCCL(e, R, R->AllGather(dC + world, dC, 1, ncclUint64, (ncclComm_t)c->comm, stream));
  std::vector<unsigned long long> h(world);
  MM_HIP(e, hipMemcpyAsync(h.data(), dC, (size_t)world * 8, hipMemcpyDeviceToHost, stream));
  MM_HIP(e, hipStreamSynchronize(stream));
  c->gatherCounts.assign(h.begin(), h.end());
  place(c);
  MM_HIP(e, c->dGathered.ensure(c->nGathered * sizeof(mm_mapping) + 64));
  MM_NCCL(e, R, R->GroupStart());
  const int rc = issue_broadcasts(c, e, R, mine, stream);
  MM_NCCL(e, R, R->GroupEnd());
  if (rc != MM_OK) return rc;
  MM_HIP(e, hipStreamSynchronize(stream));
  return MM_OK;
}

}  // namespace

void mm_comm_release(mm_ctx* c) {
  if (c->comm && g_rccl.h) (void)g_rccl.CommDestroy((ncclComm_t)c->comm);
  c->comm = nullptr; c->commWorld = 0; c->commRank = 0; c->commCopy = false;
}

extern "C" {

int mm_comm_unique_id(void* id) {
  std::string err;
  Rccl* R = rccl_open(err);
  if (!R || !id) return MM_ERR_DEVICE;
  ncclUniqueId u;
  if (R->GetUniqueId(&u) != ncclSuccess) return MM_ERR_DEVICE;
  static_assert(sizeof(ncclUniqueId) == MM_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id, &u, sizeof u);
  return MM_OK;
}

int mm_comm_init_rank(mm_ctx* c, const void* id, int rank, int world) {
  if (!id || world < 1 || rank < 0 || rank >= world) { c->err = "mm_comm_init_rank: bad argument"; return MM_ERR_ARG; }
  Rccl* R = rccl_open(c->err);
  if (!R) return MM_ERR_DEVICE;
  MM_HIP(c, hipSetDevice(c->device));
  mm_comm_release(c);
  ncclUniqueId u; std::memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  MM_NCCL(c, R, R->CommInitRank(&comm, world, u, rank));
  c->comm = comm; c->commRank = rank; c->commWorld = world; c->commCopy = false;
  MM_HIP(c, c->dCommCounts.ensure((size_t)(world + 1) * 8));
  return MM_OK;
}

int mm_comm_init_local(mm_ctx** ctxs, int n) {
  if (!ctxs || n < 1) return MM_ERR_ARG;
  mm_ctx* c0 = ctxs[0];
  std::vector<int> devs(n);
  bool distinct = true;
  for (int i = 0; i < n; i++) { devs[i] = ctxs[i]->device; for (int j = 0; j < i; j++) if (devs[j] == devs[i]) distinct = false; }
  for (int i = 0; i < n; i++) { mm_comm_release(ctxs[i]); ctxs[i]->commRank = i; ctxs[i]->commWorld = n; ctxs[i]->commCopy = true; }
  // every context starts on the copy path (device / peer copies: always available inside one process); distinct GPUs move to RCCL
  // broadcasts when a communicator can be had.  No RCCL, or a failing ncclCommInitAll, is not an error here -- the group stays on
  // peer copies, with a warning -- unless MASHMAP_HIP_REQUIRE_RCCL is set.
  if (distinct && n > 1 && !getenv("MASHMAP_HIP_NO_RCCL")) {
    std::string why;
    Rccl* R = rccl_open(why);
    std::vector<ncclComm_t> comms(n, nullptr);
    bool ok = R != nullptr;
    if (ok) { const ncclResult_t r = R->CommInitAll(comms.data(), n, devs.data()); if (r != ncclSuccess) { ok = false; why = std::string("ncclCommInitAll: ") + R->GetErrorString(r); } }
    if (ok) for (int i = 0; i < n; i++) { ctxs[i]->comm = comms[i]; ctxs[i]->commCopy = false; }
    else {
      if (getenv("MASHMAP_HIP_REQUIRE_RCCL")) { c0->err = why; return MM_ERR_DEVICE; }
      fprintf(stderr, "[mm] warning: no RCCL communicator for the local group (%s); candidate mappings are exchanged by peer copies\n", why.c_str());
    }
  }
  return MM_OK;
}

int mm_comm_info(const mm_ctx* c, int* worldSeen, char* libraryPath, size_t cap) {
  if (!c->commWorld) return MM_ERR_STATE;
  int n = c->commCopy ? c->commWorld : -1;             // contexts sharing a device exchange by copies: no communicator to ask
  if (!c->commCopy && c->comm && g_rccl.h && g_rccl.CommCount) { int k = 0; if (g_rccl.CommCount((ncclComm_t)c->comm, &k) == ncclSuccess) n = k; }
  if (worldSeen) *worldSeen = n;
  if (libraryPath && cap) { const std::string& p = c->commCopy ? std::string("(device copies, no RCCL)") : g_rccl.path; snprintf(libraryPath, cap, "%s", p.c_str()); }
  return MM_OK;
}

int mm_comm_world(const mm_ctx* c, int* rank, int* world) {
  if (rank) *rank = c->commRank;
  if (world) *world = c->commWorld;
  return c->commWorld ? MM_OK : MM_ERR_STATE;
}

int mm_allgatherv_mappings(mm_ctx* c) {
  if (!c->comm || c->commCopy) { c->err = "mm_allgatherv_mappings: mm_comm_init_rank first"; return MM_ERR_STATE; }
  if (!c->mapped || !c->haveReplayTables) { c->err = "mm_allgatherv_mappings: no candidate mappings resident"; return MM_ERR_STATE; }
  if (c->gatherThread.joinable()) { c->err = "mm_allgatherv_mappings: an overlapped exchange is in flight (mm_allgatherv_mappings_end first)"; return MM_ERR_STATE; }
  Rccl* R = rccl_open(c->err);
  if (!R) return MM_ERR_DEVICE;
  c->gathered = false;
  const int rc = exchange(c, c, R, c->dMappings.p, c->nMappings, c->stream);
  if (rc != MM_OK) return rc;
  c->gathered = true;
  return MM_OK;
}

int mm_allgatherv_mappings_begin(mm_ctx* c) {
  if (!c->comm || c->commCopy) { c->err = "mm_allgatherv_mappings_begin: mm_comm_init_rank first"; return MM_ERR_STATE; }
  if (!c->mapped || !c->haveReplayTables) { c->err = "mm_allgatherv_mappings_begin: no candidate mappings resident"; return MM_ERR_STATE; }
  if (c->gatherThread.joinable()) { c->err = "mm_allgatherv_mappings_begin: the previous exchange has not been ended"; return MM_ERR_STATE; }
  Rccl* R = rccl_open(c->err);
  if (!R) return MM_ERR_DEVICE;
  MM_HIP(c, hipSetDevice(c->device));
  if (!c->commStream) MM_HIP(c, hipStreamCreateWithFlags(&c->commStream, hipStreamNonBlocking));
  // snapshot: the next mm_map_fragments overwrites dMappings while the exchange is in flight
  const size_t n = c->nMappings;
  MM_HIP(c, c->dGatherSrc.ensure(n * sizeof(mm_mapping) + 64));
  if (n) MM_HIP(c, hipMemcpyAsync(c->dGatherSrc.p, c->dMappings.p, n * sizeof(mm_mapping), hipMemcpyDeviceToDevice, c->stream));
  MM_HIP(c, hipStreamSynchronize(c->stream));
  c->gathered = false;
  c->gatherRc = MM_OK;
  c->gatherThread = std::thread([c, R, n]() {
    ErrSink e;
    c->gatherRc = exchange(c, &e, R, c->dGatherSrc.p, n, c->commStream);
    c->gatherErr = e.err;
  });
  return MM_OK;
}

int mm_allgatherv_mappings_end(mm_ctx* c) {
  if (!c->gatherThread.joinable()) { c->err = "mm_allgatherv_mappings_end: no exchange in flight"; return MM_ERR_STATE; }
  c->gatherThread.join();
  if (c->gatherRc != MM_OK) { c->err = c->gatherErr; return c->gatherRc; }
  c->gathered = true;
  return MM_OK;
}

int mm_allgatherv_mappings_local(mm_ctx** ctxs, int n) {
  if (!ctxs || n < 1) return MM_ERR_ARG;
  mm_ctx* c0 = ctxs[0];
  for (int i = 0; i < n; i++) {
    mm_ctx* c = ctxs[i];
    if (c->commWorld != n || c->commRank != i) { c0->err = "mm_allgatherv_mappings_local: contexts are not the group of mm_comm_init_local"; return MM_ERR_STATE; }
    if (!c->mapped || !c->haveReplayTables) { c0->err = "mm_allgatherv_mappings_local: a context has no candidate mappings resident"; return MM_ERR_STATE; }
  }
  for (int i = 0; i < n; i++) {
    mm_ctx* c = ctxs[i];
    c->gatherCounts.resize(n);
    for (int r = 0; r < n; r++) c->gatherCounts[r] = ctxs[r]->nMappings;
    place(c);
    MM_HIP(c, hipSetDevice(c->device));
    MM_HIP(c, c->dGathered.ensure(c->nGathered * sizeof(mm_mapping) + 64));
  }
  if (!c0->commCopy) {
    Rccl* R = rccl_open(c0->err);
    if (!R) return MM_ERR_DEVICE;
    MM_NCCL(c0, R, R->GroupStart());
    int rc = MM_OK;
    for (int i = 0; i < n && rc == MM_OK; i++) { (void)hipSetDevice(ctxs[i]->device); rc = issue_broadcasts(ctxs[i], ctxs[i], R, ctxs[i]->dMappings.p, ctxs[i]->stream); if (rc != MM_OK) c0->err = ctxs[i]->err; }
    if (rc != MM_OK) {
      // Everything that can fail on our side (counts, slots, buffers) was settled before ncclGroupStart, so this is RCCL refusing to
      // enqueue a broadcast.  The group is closed first -- aborting a communicator inside an open group is not a supported sequence --
      // then the communicators are aborted and the contexts fall back to peer copies for whatever comes next.  Without ncclCommAbort
      // in the bound library the communicators cannot be torn down safely: that is reported as such, not papered over.
      std::string keep = c0->err;
      (void)R->GroupEnd();
      for (int i = 0; i < n; i++) {
        if (ctxs[i]->comm && R->CommAbort) (void)R->CommAbort((ncclComm_t)ctxs[i]->comm);
        ctxs[i]->comm = nullptr; ctxs[i]->commCopy = true;
      }
      c0->err = keep + (R->CommAbort ? "" : " (the bound RCCL has no ncclCommAbort: its communicators were dropped, not destroyed)");
      return R->CommAbort ? rc : MM_ERR_DEVICE;
    }
    MM_NCCL(c0, R, R->GroupEnd());
  } else {
    // contexts sharing a device: slot r of every context is a device copy of context r's records (mm_map_fragments has synchronised
    // every source stream)
    for (int i = 0; i < n; i++) {
      mm_ctx* c = ctxs[i];
      MM_HIP(c, hipSetDevice(c->device));
      for (int r = 0; r < n; r++) {
        const size_t bytes = c->gatherCounts[r] * sizeof(mm_mapping);
        if (!bytes) continue;
        char* slot = (char*)c->dGathered.p + c->gatherDisp[r] * sizeof(mm_mapping);
        if (ctxs[r]->device == c->device) MM_HIP(c, hipMemcpyAsync(slot, ctxs[r]->dMappings.p, bytes, hipMemcpyDeviceToDevice, c->stream));
        else MM_HIP(c, hipMemcpyPeerAsync(slot, c->device, ctxs[r]->dMappings.p, ctxs[r]->device, bytes, c->stream));
      }
    }
  }
  for (int i = 0; i < n; i++) { mm_ctx* c = ctxs[i]; MM_HIP(c, hipSetDevice(c->device)); MM_HIP(c, hipStreamSynchronize(c->stream)); c->gathered = true; }
  return MM_OK;
}

int mm_gathered_counts(const mm_ctx* c, size_t* perRank, size_t* total) {
  if (!c->gathered) return MM_ERR_STATE;
  if (perRank) for (size_t r = 0; r < c->gatherCounts.size(); r++) perRank[r] = c->gatherCounts[r];
  if (total) *total = c->nGathered;
  return MM_OK;
}

int mm_gathered_download(mm_ctx* c, mm_mapping* out, size_t cap) {
  if (!c->gathered) { c->err = "mm_gathered_download: nothing gathered"; return MM_ERR_STATE; }
  if (c->nGathered > cap) { c->err = "mm_gathered_download: destination too small"; return MM_ERR_ARG; }
  MM_HIP(c, hipSetDevice(c->device));
  if (c->nGathered) {
    MM_HIP(c, hipMemcpyAsync(out, c->dGathered.p, c->nGathered * sizeof(mm_mapping), hipMemcpyDeviceToHost, c->stream));
    MM_HIP(c, hipStreamSynchronize(c->stream));
  }
  return MM_OK;
}

int mm_gathered_device(const mm_ctx* c, const mm_mapping** d, size_t* total) {
  if (!c->gathered) return MM_ERR_STATE;
  if (d) *d = c->dGathered.as<mm_mapping>();
  if (total) *total = c->nGathered;
  return MM_OK;
}

// replica of src's resident index on dst's GPU (device-to-device over xGMI; no second build, no host round trip).  dst gets the
// device index only: the host mirrors behind mm_index_download stay with src.
int mm_index_replicate(mm_ctx* dst, mm_ctx* src) {
  if (!src->idx.ready) { dst->err = "mm_index_replicate: the source context has no index"; return MM_ERR_STATE; }
  if (dst->P.kmerSize != src->P.kmerSize || dst->P.segLength != src->P.segLength || dst->P.sketchSize != src->P.sketchSize) {
    dst->err = "mm_index_replicate: contexts differ in kmerSize / segLength / sketchSize"; return MM_ERR_ARG;
  }
  MM_HIP(dst, hipSetDevice(src->device));
  MM_HIP(dst, hipStreamSynchronize(src->stream));
  MM_HIP(dst, hipSetDevice(dst->device));
  DeviceIndex& D = dst->idx; DeviceIndex& S = src->idx;
  D.ready = false;
  DevBuf* d[] = {&D.evKey, &D.evAux, &D.evHash, &D.contigOff, &D.opKey, &D.opAux, &D.opHash, &D.blockOff, &D.evBlock, &D.contigBlock, &D.contigLen, &D.refGroup, &D.htSlots, &D.htTags, &D.filter, &D.ptKeys};
  DevBuf* s[] = {&S.evKey, &S.evAux, &S.evHash, &S.contigOff, &S.opKey, &S.opAux, &S.opHash, &S.blockOff, &S.evBlock, &S.contigBlock, &S.contigLen, &S.refGroup, &S.htSlots, &S.htTags, &S.filter, &S.ptKeys};
  for (size_t i = 0; i < sizeof d / sizeof d[0]; i++) {
    if (!s[i]->bytes) continue;
    MM_HIP(dst, d[i]->ensure(s[i]->bytes));
    if (dst->device == src->device) MM_HIP(dst, hipMemcpyAsync(d[i]->p, s[i]->p, s[i]->bytes, hipMemcpyDeviceToDevice, dst->stream));
    else MM_HIP(dst, hipMemcpyPeerAsync(d[i]->p, dst->device, s[i]->p, src->device, s[i]->bytes, dst->stream));
  }
  MM_HIP(dst, hipStreamSynchronize(dst->stream));
  D.nRec = S.nRec; D.nKeys = S.nKeys; D.nPoints = S.nPoints; D.nContigs = S.nContigs; D.htCap = S.htCap; D.nOpen = S.nOpen; D.filterMask = S.filterMask; D.tagged = S.tagged;
  D.ready = true;
  dst->freqThreshold = src->freqThreshold;
  dst->mapped = false;
  return MM_OK;
}

}  // extern "C"
